// stft_block.h — one WORKGROUP per frame: the packed real FFT of the long transforms, n_fft = 8192 (Nc = 4096),
// 16384 (Nc = 8192), 32768 (Nc = 16384: round 3, the largest whose frame fits the CU's LDS as complex slots) and 65536
// (Nc = 32768, round 5: PLANAR exchanges — the real parts of an exchange go through the LDS first, then the imaginary parts,
// 139 KB at a time), which do not fit one wave's registers (stft_wave.h stops at n_fft = 4096).
//
// T = Nc / 16 threads (256, 512 or 1024) work on one frame, 16 complex points per thread.  Stockham autosort, decimation
// in time, radix-16 passes in registers with an LDS exchange (and a workgroup barrier) between passes:
//
//     Nc = 4096:          16 (Ns = 1)  x 16 (Ns = 16)  x 16 (Ns = 256)
//     Nc = 8192:  2 (Ns = 1) x 16 (Ns = 2)  x 16 (Ns = 32)  x 16 (Ns = 512)
//     Nc = 16384: 4 (Ns = 1) x 16 (Ns = 4)  x 16 (Ns = 64)  x 16 (Ns = 1024)
//     Nc = 32768: 8 (Ns = 1) x 16 (Ns = 8)  x 16 (Ns = 128) x 16 (Ns = 2048)
//
// A pass with sub-transform size Ns: thread t owns butterfly j = t, k = t mod Ns,
//     v_r = in[t + T r] * W_{16 Ns}^{r k}   (r = 0..15),     out[(t - k) 16 + k + Ns c] = DFT16(v)_c
// (the radix-2 first pass of Nc = 8192: butterflies j = t + T m', m' < 8, on the points the thread loaded itself,
// out[2 j + r]; the radix-4 first pass of Nc = 16384 likewise: j = t + T m', m' < 4, slots m', m' + 4, m' + 8, m' + 12).  The twiddled passes run as the fused-multiply-add butterflies of stft_wave.h (bfly4_tw): 10 per-thread
// constants per pass, held in registers for the whole launch.  After the last pass thread t holds Z[t + T c], c = 0..15;
// Z goes through LDS once more so that every thread can read the mirror partners Z[Nc - k] of its bins k = t + T c, c < 8,
// and emit X[k] and X[Nc - k] of the real-FFT split pass together.
//
// LDS layout per exchange (slot = one cf32): after the Ns <= 2 pass pad1 (two pad slots per 32: a thread writes 16 / 2
// consecutive slots), after the Ns = 16 pass i + 16 (i >> 8) (the four 16-lane groups of a wave land on different bank
// halves), after the Ns = 32 pass and for Z the identity.  Every read is "in[t + T r]": per-thread base + immediate.
//
// Like stft_wave.h this header compiles for gfx950 (hipcc) and for the CPU lane emulator in tests/emu (g++).
#pragma once
#include "stft_wave.h"

namespace th {

template <int LOG2_NC>
struct BlockFft {
    static_assert(LOG2_NC >= 12 && LOG2_NC <= 15, "n_fft = 8192, 16384, 32768 or 65536");
    static constexpr int NC = 1 << LOG2_NC;
    static constexpr int T = NC / 16;           // (virtual) threads per frame
    static constexpr int FIRST_R = LOG2_NC == 12 ? 16 : LOG2_NC == 13 ? 2 : LOG2_NC == 14 ? 4 : 8;  // radix of the untwiddled first pass
    // PLANAR: an exchange image of Nc complex slots (256 KB at Nc = 32768) does not fit the CU's 160 KB of LDS; the buffer then
    // holds Nc FLOATS and every exchange runs twice, PART 0 = real parts, PART 1 = imaginary parts (the accessors below take
    // PART as a template argument; PART = -1 is the complex-slot form of the smaller sizes)
    static constexpr bool PLANAR = LOG2_NC == 15;
    static constexpr bool R2_FIRST = FIRST_R != 16;  // (a small first pass, then THREE twiddled radix-16 passes)
    static constexpr int NS_A = R2_FIRST ? FIRST_R : 1;  // sub-transform sizes of the three radix-16 passes
    static constexpr int NS_B = 16 * NS_A, NS_C = 16 * NS_B;
    static_assert(16 * NS_C == NC, "last pass completes the transform");
    static constexpr int NTW = 10;                       // FMA-plan constants per twiddled pass (WaveFft::FMA_TW)
    static constexpr int BUF_LEN = NC + NC / 16 + 2;     // cf32 slots: padded image + the copy of Z[0] at slot Nc

    template <int NS>
    static TH_HD uint32_t pad(uint32_t i) {
        if constexpr (NS <= 2) return pad1(i);
        else if constexpr (NS == 16) return i + 16u * (i >> 8);
        // (planar, after the Ns = 8 pass: a thread's outputs are 8 floats apart and the threads of a wave 128 — four of a half-wave's
        // eight thread groups on the same banks; 8 floats of skew per 128: linear in r for in[t + T r], T a multiple of 128)
        else if constexpr (PLANAR && NS == 8) return i + 8u * (i >> 7);
        else return i;
    }
    // slot i of the exchange buffer: a complex slot (PART < 0) or the float of one part
    template <int PART>
    static TH_HD void put(cf32 *buf, uint32_t i, const cf32 &v) {
        if constexpr (PART < 0) buf[i] = v;
        else reinterpret_cast<float *>(buf)[i] = PART ? v.im : v.re;
    }
    template <int PART>
    static TH_HD void get(const cf32 *buf, uint32_t i, cf32 &v) {
        if constexpr (PART < 0) v = buf[i];
        else if constexpr (PART == 0) v.re = reinterpret_cast<const float *>(buf)[i];
        else v.im = reinterpret_cast<const float *>(buf)[i];
    }
    // pad<NS>(t + T r) = pad<NS>(t) + PSTEP<NS> r  (T is a multiple of 256)
    template <int NS>
    static constexpr uint32_t pstep() {
        return NS <= 2 ? (uint32_t)(T + T / 16) : NS == 16 ? (uint32_t)(T + 16 * (T / 256)) : (PLANAR && NS == 8) ? (uint32_t)(T + T / 16) : (uint32_t)T;
    }

    // the 10 constants of a twiddled pass for twiddle index k: w = W_{16 Ns}^k = tw[k S], S = 2 Nc / (16 Ns)
    // (tw[i] = exp(-2 pi i * i / n_fft), n_fft = 2 Nc): w^4, w^8, then t, t^2 for t = w W16^m', m' = 0..3
    template <int NS>
    static TH_HD void load_tw(uint32_t t, cf32 (&w)[NTW], const cf32 *tw) {
        constexpr uint32_t S = 2 * NC / (16 * NS), S16 = 2 * NC / 16, M = 2 * NC;
        const uint32_t k = t & (uint32_t)(NS - 1);
        TH_UNROLL for (int e = 0; e < NTW; e++) {
            const uint32_t mp = e >= 2 ? (uint32_t)(e - 2) / 2 : 0u, pw = e < 2 ? 4u * (uint32_t)(e + 1) : 1u + ((uint32_t)(e - 2) & 1u);
            w[e] = tw[(pw * k * S + (e >= 2 ? pw * mp * S16 : 0u)) % M];
        }
    }

    // twiddled radix-16 butterfly in place (WaveFft<10>::dft16_tw_to_planes without the stores): natural output c ends
    // up in v[slot_tw(c)]
    static TH_HD void dft16_tw(cf32 (&v)[16], const cf32 (&w)[NTW]) {
        TH_UNROLL for (int b = 0; b < 4; b++) bfly4_tw(v[b], v[4 + b], v[8 + b], v[12 + b], w[0], w[1]);
        bfly4_tw(v[0], v[1], v[2], v[3], w[2], w[3]);      // m' = 0: inputs y_b[0] in v[b]
        bfly4_tw(v[8], v[9], v[10], v[11], w[4], w[5]);    // m' = 1: y_b[1] in v[8 + b]
        bfly4_tw(v[4], v[5], v[6], v[7], w[6], w[7]);      // m' = 2: y_b[2] in v[4 + b]
        bfly4_tw(v[12], v[13], v[14], v[15], w[8], w[9]);  // m' = 3
    }
    // output c = m' + 4 m'' of dft16_tw: group base 4 pi(m'), pi = (0, 2, 1, 3); inside a group (X0, X2, X1, X3)
    static TH_HD constexpr int slot_tw(int c) {
        const int mp = c & 3, mpp = c >> 2;
        const int base = 4 * (mp == 1 ? 2 : mp == 2 ? 1 : mp);
        return base + (mpp == 1 ? 2 : mpp == 2 ? 1 : mpp);
    }

    // ---- first pass(es): z[m] = windowed point t + T m of the frame -> LDS (exchange 1); compute and store separately,
    // so that a planar exchange can store the two parts one after the other
    static TH_HD void pass_first_compute(cf32 (&z)[16]) {
        if constexpr (FIRST_R == 2) {
            // radix 2, Ns = 1: butterfly j = t + T m' pairs the points j and j + Nc/2 = slots m', m' + 8
            TH_UNROLL for (int m = 0; m < 8; m++) fft2(z[m], z[m + 8]);
        } else if constexpr (FIRST_R == 4) {
            // radix 4, Ns = 1: butterfly j = t + T m' on the points j + r Nc/4 = slots m' + 4 r
            TH_UNROLL for (int m = 0; m < 4; m++) fft4(z[m], z[m + 4], z[m + 8], z[m + 12]);
        } else if constexpr (FIRST_R == 8) {
            // radix 8, Ns = 1: butterfly j = t + T m' on the points j + r Nc/8 = slots m' + 2 r; X[c] lands in slot m' + 2 dft8_slot(c)
            TH_UNROLL for (int m = 0; m < 2; m++) {
                cf32 v[8];
                TH_UNROLL for (int r = 0; r < 8; r++) v[r] = z[m + 2 * r];
                dft8(v);
                TH_UNROLL for (int r = 0; r < 8; r++) z[m + 2 * r] = v[r];
            }
        } else {
            dft16(z);  // radix 16, Ns = 1: butterfly j = t on the thread's own 16 points
        }
    }
    template <int PART = -1>
    static TH_HD void pass_first_store(uint32_t t, const cf32 (&z)[16], cf32 *buf) {
        if constexpr (FIRST_R == 2) {  // out[2 j + r]
            TH_UNROLL for (int m = 0; m < 8; m++) {
                const uint32_t o = 2u * (t + (uint32_t)T * m);
                put<PART>(buf, o, z[m]);
                put<PART>(buf, o + 1, z[m + 8]);
            }
        } else if constexpr (FIRST_R == 4) {  // out[4 j + r]
            TH_UNROLL for (int m = 0; m < 4; m++) {
                const uint32_t o = 4u * (t + (uint32_t)T * m);
                put<PART>(buf, o, z[m]);
                put<PART>(buf, o + 1, z[m + 4]);
                put<PART>(buf, o + 2, z[m + 8]);
                put<PART>(buf, o + 3, z[m + 12]);
            }
        } else if constexpr (FIRST_R == 8) {  // out[8 j + c]: eight consecutive slots (planar: 32 bytes, two 16-byte stores)
            TH_UNROLL for (int m = 0; m < 2; m++) {
                const uint32_t o = 8u * (t + (uint32_t)T * m);
                if constexpr (PART >= 0) {
                    float *const q = reinterpret_cast<float *>(buf) + o;
                    TH_UNROLL for (int h = 0; h < 2; h++) {
                        f32x4 w;
                        w.a = PART ? z[m + 2 * dft8_slot(4 * h + 0)].im : z[m + 2 * dft8_slot(4 * h + 0)].re;
                        w.b = PART ? z[m + 2 * dft8_slot(4 * h + 1)].im : z[m + 2 * dft8_slot(4 * h + 1)].re;
                        w.c = PART ? z[m + 2 * dft8_slot(4 * h + 2)].im : z[m + 2 * dft8_slot(4 * h + 2)].re;
                        w.d = PART ? z[m + 2 * dft8_slot(4 * h + 3)].im : z[m + 2 * dft8_slot(4 * h + 3)].re;
                        lds_st4(q + 4 * h, w);
                    }
                } else {
                    TH_UNROLL for (int c = 0; c < 8; c++) put<PART>(buf, o + (uint32_t)c, z[m + 2 * dft8_slot(c)]);
                }
            }
        } else {  // out[16 t + c]
            TH_UNROLL for (int c = 0; c < 16; c++) put<PART>(buf, pad1(16u * t) + c, z[dft16_slot(c)]);
        }
    }
    static TH_HD void pass_first(uint32_t t, cf32 (&z)[16], cf32 *buf) {
        pass_first_compute(z);
        pass_first_store<-1>(t, z, buf);
    }
    // layout tag of the exchange the first pass writes (see pad<>): pad1 after the radix-16 first pass, linear after the radix-2 one
    static constexpr int FIRST_LAYOUT = R2_FIRST ? 64 : 1;
    // read in[t + T r] of the exchange written by the pass with sub-size NS_PREV (layout pad<NS_PREV>)
    template <int NS_PREV, int PART = -1>
    static TH_HD void read_in(uint32_t t, cf32 (&z)[16], const cf32 *buf) {
        const uint32_t b = pad<NS_PREV>(t);
        TH_UNROLL for (int r = 0; r < 16; r++) get<PART>(buf, b + pstep<NS_PREV>() * (uint32_t)r, z[r]);
    }
    // twiddled radix-16 pass with sub-size NS (not the last): registers -> LDS, out[(t - k) 16 + k + NS c]
    template <int NS>
    static TH_HD void pass_mid_compute(cf32 (&z)[16], const cf32 (&w)[NTW]) { dft16_tw(z, w); }
    template <int NS, int PART = -1>
    static TH_HD void pass_mid_store(uint32_t t, const cf32 (&z)[16], cf32 *buf) {
        const uint32_t k = t & (uint32_t)(NS - 1), o = (t - k) * 16u + k;
        TH_UNROLL for (int c = 0; c < 16; c++) put<PART>(buf, pad<NS>(o + (uint32_t)(NS * c)), z[slot_tw(c)]);
    }
    // last pass: z <- Z[t + T c] in natural slot order
    static TH_HD void pass_last(cf32 (&z)[16], const cf32 (&w)[NTW]) {
        dft16_tw(z, w);
        cf32 o[16];
        TH_UNROLL for (int c = 0; c < 16; c++) o[c] = z[slot_tw(c)];
        TH_UNROLL for (int c = 0; c < 16; c++) z[c] = o[c];
    }
    // publish Z (natural order, plus Z[0] again at slot Nc so that the mirror of bin 0 needs no wrap-around)
    template <int PART = -1>
    static TH_HD void write_z(uint32_t t, const cf32 (&z)[16], cf32 *buf) {
        TH_UNROLL for (int c = 0; c < 16; c++) put<PART>(buf, t + (uint32_t)T * c, z[c]);
        if (t == 0) put<PART>(buf, (uint32_t)NC, z[0]);
    }
    // Split pass: thread t owns the pairs (k, Nc - k), k = t + T c, c < 8, and — thread 0 — the self-mirrored bin Nc/2.
    // stw_t = W_{n_fft}^t; W^{t + T c} = stw_t * omega^c, omega = exp(-i pi / 16).  emit(bin, |X[bin]|^2).
    // (in two steps, so that the kernel can request the next frame's samples between the LDS reads and the arithmetic)
    template <int PART = -1>
    static TH_HD void split_read(uint32_t t, const cf32 *buf, cf32 (&zm)[8]) {
        TH_UNROLL for (int c = 0; c < 8; c++) get<PART>(buf, (uint32_t)NC - t - (uint32_t)T * c, zm[c]);
    }
    template <class Emit>
    static TH_HD void split(uint32_t t, const cf32 (&z)[16], const cf32 *buf, cf32 stw_t, Emit emit) {
        cf32 zm[8];
        split_read(t, buf, zm);
        split_compute(t, z, zm, stw_t, emit);
    }
    template <class Emit>
    static TH_HD void split_compute(uint32_t t, const cf32 (&z)[16], const cf32 (&zm)[8], cf32 stw_t, Emit emit) {
        constexpr float OC[9] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                                 0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f,
                                 0.19509032201612826785f, 0.0f};  // cos(pi c / 16); sin(pi c / 16) = OC[8 - c]
        TH_UNROLL for (int c = 0; c < 8; c++) {
            const cf32 w = cmul_c(stw_t, OC[c], -OC[8 - c]);
            const cf32 zk = z[c];
            const float er = zk.re + zm[c].re, ei = zk.im - zm[c].im;
            const float dr = zk.re - zm[c].re, di = zk.im + zm[c].im;
            const float xr = th_fma(di, w.re, th_fma(dr, w.im, er)), xi = th_fma(di, w.im, th_fma(-dr, w.re, ei));
            const float yr = th_fma(2.0f, er, -xr), yi = th_fma(2.0f, ei, -xi);
            emit(t + (uint32_t)T * c, xr * xr + xi * xi);
            emit((uint32_t)NC - t - (uint32_t)T * c, yr * yr + yi * yi);
        }
        if (t == 0) {  // bin Nc/2 = Z[8 T] of thread 0: its own mirror, W^(Nc/2) = -i
            const cf32 zh = z[8];
            const float xr = 2.0f * zh.re, xi = -2.0f * zh.im;  // e = (2 re, 0), d = (0, 2 im): X = e + (-i)(-i d) ... = (2 re, -2 im)
            emit((uint32_t)NC / 2, xr * xr + xi * xi);
        }
    }
};

}  // namespace th
