// stft_wave_multi.h — several SHORT frames per wavefront (n_fft = 1024: two, n_fft = 512: four).
//
// The one-frame-per-wave plan of stft_wave.h is built around 16 complex points per lane (n_fft = 2048).  A 512- or
// 256-point packed FFT spread over 64 lanes leaves 8 or 4 points per lane: more, smaller passes and one LDS exchange
// more per frame (n_fft = 1024 ran at 0.39 of the HBM roofline, n_fft = 512 had no register kernel at all and fell to
// the generic kernel at 0.05).  Here the 64 lanes are split into G = 1024 / Nc groups of L = 64 / G lanes, every group
// owns one frame, and every lane still holds P = 16 points:
//
//     Nc = 16 (pass 1, in registers) x 16 (pass 2, in registers) x R3,   R3 = Nc / 256 = 2 or 1
//
// so the whole wave executes the very instruction stream of the 2048-point plan — the same two radix-16 passes, the
// same two plane exchanges (ds_write_addtid_b32 planes over all 64 lanes at once, 16-byte reads), the same mirror-local
// split — on G frames at a time; only the last pass shrinks (radix 2 for Nc = 512, none for Nc = 256).
//
// Index maps (l = lane within its group, g = group):
//   pass 1: lane l owns column col(l) = S (l & 15) + (l >> 4) of its frame, S = L / 16: complex points col + L m
//   pass 2: lane l = S c + a owns butterfly j = 16 a + c (twiddle index c); inputs in[j + L r] = pass-1 output c of column
//           S r + a = plane c, lanes g L + 16 a + r: 16 consecutive dwords
//   pass 3: lane l owns the mirror pairs A_q = L q + l, B_q = 256 - A_q (q < 8 / R3; lane 0, q = 0: B = 128); input
//           in[jj + 256 r] = pass-2 output jj >> 4 of butterfly 16 r + (jj & 15) = plane jj >> 4, lanes g L + R3 (jj & 15) + r
//   bins:   k = jj + 256 s (s < R3) and Nc - k from the same lane's registers (real-FFT split pass)
//
// Like stft_wave.h this header also compiles for the CPU lane emulator in tests/emu.
#pragma once
#include "stft_wave.h"

namespace th {

template <int LOG2_NC>
struct WaveFftM {
    static_assert(LOG2_NC == 8 || LOG2_NC == 9, "n_fft = 512 or 1024");
    static constexpr int NC = 1 << LOG2_NC;
    static constexpr int G = 1024 / NC;   // frames per wave
    static constexpr int L = 64 / G;      // lanes per frame
    static constexpr int P = 16;          // complex points per lane
    static constexpr int S = L / 16;
    static constexpr int R3 = NC / 256;
    static constexpr int NS3 = 256;
    static constexpr int NQ = 8 / R3;     // mirror pairs of butterflies per lane
    static constexpr int PITCH1 = 68, PITCH2 = 64;  // dwords per plane (2-way conflicts at worst on the 16-/8-byte reads)
    static constexpr int SLAB_LEN = 16 * PITCH1;    // cf32 units: 32 planes of PITCH1 dwords (exchange 1 is the larger)
    // pass 2 runs as the fused-multiply-add butterflies of the 2048 plan (WaveFft<10>::dft16_tw_to_planes): 10 table
    // entries per twiddle index c — w^4, w^8, then t, t^2 for t = w W16^m', m' = 0..3, w = W_256^c
    static constexpr int NT2 = 10;
    static constexpr int T2_LEN = NT2 * 16, T3_LEN = (R3 - 1) * NS3;
    // tw[i] = exp(-2 pi i * i / n_fft), n_fft = 2 Nc
    static TH_HD void fill_tables(uint32_t tid, uint32_t nthr, const cf32 *tw, cf32 *t2, cf32 *t3) {
        constexpr uint32_t S = 2 * NC / 256, S16 = 2 * NC / 16, M = 2 * NC;
        for (uint32_t i = tid; i < (uint32_t)T2_LEN; i += nthr) {
            const uint32_t e = i / 16, k = i % 16;
            const uint32_t mp = e >= 2 ? (e - 2) / 2 : 0, pw = e < 2 ? 4u * (e + 1u) : 1u + ((e - 2) & 1u);
            t2[i] = tw[(pw * k * S + (e >= 2 ? pw * mp * S16 : 0u)) % M];
        }
        for (uint32_t i = tid; i < (uint32_t)T3_LEN; i += nthr) t3[i] = tw[2 * i];  // W_Nc^jj (R3 = 2)
    }
    static TH_HD uint32_t grp(uint32_t lane) { return lane / L; }
    static TH_HD uint32_t lig(uint32_t lane) { return lane % L; }
    static TH_HD uint32_t lane_col(uint32_t lane) {
        const uint32_t l = lig(lane);
        return (uint32_t)S * (l & 15u) + (l >> 4);
    }

    static TH_HD void pass1(uint32_t lane, cf32 (&z)[P], cf32 *slab) { WaveFft<10>::template dft16_to_planes<PITCH1>(lane, z, slab); }
    static TH_HD void read1(uint32_t lane, cf32 (&z)[P], const cf32 *slab) {
        const uint32_t l = lig(lane), a = l % S, c = l / S;
        const float *const sf = reinterpret_cast<const float *>(slab) + c * PITCH1 + grp(lane) * L + 16u * a;
        f32x4 re[4], im[4];
        TH_UNROLL for (int t = 0; t < 4; t++) {
            re[t] = lds_ld4(sf + 4 * t);
            im[t] = lds_ld4(sf + 16 * PITCH1 + 4 * t);
        }
        TH_UNROLL for (int t = 0; t < 4; t++) {
            z[4 * t] = {re[t].a, im[t].a};
            z[4 * t + 1] = {re[t].b, im[t].b};
            z[4 * t + 2] = {re[t].c, im[t].c};
            z[4 * t + 3] = {re[t].d, im[t].d};
        }
    }
    static TH_HD void load_t2(uint32_t lane, cf32 (&w2)[NT2], const cf32 *t2) {
        const uint32_t c = lig(lane) / S;
        TH_UNROLL for (int r = 0; r < NT2; r++) w2[r] = lds_ld(&t2[r * 16 + c]);
    }
    // twiddles + radix-16 butterfly + stores into the planes of exchange 2
    static TH_HD void pass2_w(uint32_t lane, cf32 (&z)[P], const cf32 (&w2)[NT2], cf32 *slab) {
        static_assert(WaveFft<10>::PITCH2 == PITCH2 && WaveFft<10>::NT2 == NT2, "shares the 2048 plan's butterfly");
        WaveFft<10>::template dft16_tw_to_planes<0>(lane, z, w2, slab);
    }

    static TH_HD uint32_t jj_a(uint32_t l, int q) { return (uint32_t)L * q + l; }
    static TH_HD uint32_t jj_b(uint32_t l, int q) { return (q == 0 && l == 0) ? (uint32_t)NS3 / 2 : (uint32_t)NS3 - (uint32_t)L * q - l; }
    // exchange-2 read: za[q][r] = in[A_q + 256 r], zb[q][r] = in[B_q + 256 r]
    static TH_HD void read2_paired(uint32_t lane, cf32 (&za)[NQ][R3], cf32 (&zb)[NQ][R3], const cf32 *slab) {
        const uint32_t l = lig(lane);
        const float *const sf = reinterpret_cast<const float *>(slab) + grp(lane) * L;
        TH_UNROLL for (int q = 0; q < NQ; q++) {
            const uint32_t a = jj_a(l, q), b = jj_b(l, q);
            const float *const pa = sf + (a >> 4) * PITCH2 + (uint32_t)R3 * (a & 15u), *const pb = sf + (b >> 4) * PITCH2 + (uint32_t)R3 * (b & 15u);
            if constexpr (R3 == 2) {  // two consecutive dwords per component: 8-byte reads
                const cf32 ar = lds_ld(reinterpret_cast<const cf32 *>(pa)), ai = lds_ld(reinterpret_cast<const cf32 *>(pa + 16 * PITCH2));
                const cf32 br = lds_ld(reinterpret_cast<const cf32 *>(pb)), bi = lds_ld(reinterpret_cast<const cf32 *>(pb + 16 * PITCH2));
                za[q][0] = {ar.re, ai.re};
                za[q][R3 - 1] = {ar.im, ai.im};
                zb[q][0] = {br.re, bi.re};
                zb[q][R3 - 1] = {br.im, bi.im};
            } else {
                za[q][0] = {lds_ldf(pa), lds_ldf(pa + 16 * PITCH2)};
                zb[q][0] = {lds_ldf(pb), lds_ldf(pb + 16 * PITCH2)};
            }
        }
    }
    static constexpr int NW3 = R3 == 2 ? NQ : 1;  // last-pass twiddles per lane and side (none for R3 = 1)
    static TH_HD void load_t3_paired(uint32_t lane, cf32 (&wa)[NW3], cf32 (&wb)[NW3], const cf32 *t3) {
        if constexpr (R3 == 2) {
            const uint32_t l = lig(lane);
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                wa[q] = lds_ld(&t3[jj_a(l, q)]);
                wb[q] = lds_ld(&t3[jj_b(l, q)]);
            }
        }
    }
    // last pass (radix 2, Nc = 512 only): (v0, v1) <- (v0 + w v1, v0 - w v1)
    static TH_HD void pass3_paired_w(cf32 (&za)[NQ][R3], cf32 (&zb)[NQ][R3], const cf32 (&wa)[NW3], const cf32 (&wb)[NW3]) {
        if constexpr (R3 == 2) {
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                bfly2_tw(za[q][0], za[q][R3 - 1], wa[q].re, wa[q].im);
                bfly2_tw(zb[q][0], zb[q][R3 - 1], wb[q].re, wb[q].im);
            }
        }
    }
    // bin index of pair (q, s): A_q + 256 s; lane 0, q = 0: butterfly 0 pairs inside itself (s < R3/2), butterfly 128
    // inside itself (the remaining s), exactly as WaveFft::split_k
    static TH_HD int32_t split_k(uint32_t l, int q, int s) {
        if constexpr (R3 == 1) return (int32_t)(L * q + l);
        const int32_t lane_lo = (int32_t)l, lane_hi = l == 0 ? -(int32_t)((R3 - 1) * NS3 / 2) : lane_lo;
        return (q == 0 && s >= R3 / 2 ? lane_hi : lane_lo) + L * q + s * NS3;
    }
    static TH_HD void load_stw_paired(uint32_t lane, cf32 (&ws)[NQ][R3], const cf32 *stw) {
        const uint32_t l = lig(lane);
        TH_UNROLL for (int q = 0; q < NQ; q++)
            TH_UNROLL for (int s = 0; s < R3; s++) ws[q][s] = lds_ld(&stw[split_k(l, q, s)]);
    }
    // Output addressing as in WaveFft::split_base: every bin index is "per-lane base + compile-time constant" with opaque
    // bases, so that a store is one instruction with an immediate offset.  k = lo + C (or hi + C' for q = 0, s >= R3/2),
    // Nc - k = mlo + (CMAX - C) (or mhi + ...), C = L q + s Ns3.
    static constexpr int CMAX = L * (NQ - 1) + (R3 - 1) * NS3;
    struct SplitBase {
        uint32_t lo, hi, mlo, mhi;
    };
    static TH_HD SplitBase split_base(uint32_t lane) {
        const uint32_t l = lig(lane), rl = l ^ (uint32_t)(L - 1);  // rl = L - 1 - l
        SplitBase b;
        b.lo = l;
        b.mlo = rl + (uint32_t)(NC - CMAX - (L - 1));
        b.hi = l == 0 ? (uint32_t)NS3 / 2 : l + (uint32_t)(R3 / 2) * NS3;
        b.mhi = l == 0 ? (uint32_t)(NC - NS3 / 2 - (R3 / 2 > 0 ? R3 / 2 - 1 : 0) * NS3) : rl + (uint32_t)(NC - (R3 - 1) * NS3 - (L - 1));
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(b.hi), "+v"(b.mlo), "+v"(b.mhi));
#endif
        return b;
    }
    // Split pass on lane-local pairs: emit(base, C, |X[base + C]|^2) once for every bin of the lane's frame that this lane owns.
    // stw[k] = exp(-2 pi i k / n_fft), w_mid = stw[Nc/2].
    template <class Emit>
    static TH_HD void split_paired_w(uint32_t lane, const cf32 (&za)[NQ][R3], const cf32 (&zb)[NQ][R3], const cf32 (&ws)[NQ][R3],
                                     cf32 w_mid, Emit emit) {
        const uint32_t l = lig(lane);
        const bool l0 = l == 0;
        const SplitBase sb = split_base(lane);
        auto pair = [&](cf32 zk, cf32 zm, cf32 w, uint32_t kb, int kc, uint32_t mb, int mc, bool both) {
            const float er = zk.re + zm.re, ei = zk.im - zm.im;
            const float dr = zk.re - zm.re, di = zk.im + zm.im;
            const float xr = th_fma(di, w.re, th_fma(dr, w.im, er)), xi = th_fma(di, w.im, th_fma(-dr, w.re, ei));
            emit(kb, kc, xr * xr + xi * xi);
            if (both) {
                const float yr = th_fma(2.0f, er, -xr), yi = th_fma(2.0f, ei, -xi);
                emit(mb, mc, yr * yr + yi * yi);
            }
        };
        if constexpr (R3 == 1) {
            // lane l: pairs (Z[L q + l], Z[256 - L q - l]); lane 0, q = 0: Z[0] against itself (bins 0 and Nc) and, from
            // zb, the self-mirrored Z[128] (bin 128 only: its "mirror" emit would be the same bin again)
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                cf32 zm = zb[q][0];
                if (q == 0) {
                    zm.re = l0 ? za[0][0].re : zm.re;
                    zm.im = l0 ? za[0][0].im : zm.im;
                }
                pair(za[q][0], zm, ws[q][0], sb.lo, L * q, sb.mlo, CMAX - L * q, true);
            }
            if (l0) pair(zb[0][0], zb[0][0], w_mid, (uint32_t)NC / 2, 0, 0u, 0, false);
        } else {
            TH_UNROLL for (int q = 0; q < NQ; q++) {
                TH_UNROLL for (int s = 0; s < R3; s++) {
                    cf32 zk = za[q][s], zm = zb[q][R3 - 1 - s];
                    if (q == 0) {
                        const int rp = s - R3 / 2;
                        const cf32 zk0 = s < R3 / 2 ? za[0][s] : zb[0][rp < 0 ? 0 : rp];
                        const cf32 zm0 = s < R3 / 2 ? za[0][(R3 - s) % R3] : zb[0][R3 - 1 - (rp < 0 ? 0 : rp)];
                        if (s >= R3 / 2) {
                            zk.re = l0 ? zk0.re : zk.re;
                            zk.im = l0 ? zk0.im : zk.im;
                        }
                        zm.re = l0 ? zm0.re : zm.re;
                        zm.im = l0 ? zm0.im : zm.im;
                    }
                    if (q == 0 && s >= R3 / 2) pair(zk, zm, ws[q][s], sb.hi, (s - R3 / 2) * NS3, sb.mhi, (R3 - 1 - s) * NS3, true);
                    else pair(zk, zm, ws[q][s], sb.lo, L * q + s * NS3, sb.mlo, CMAX - (L * q + s * NS3), true);
                }
            }
            if (l0) {  // the self-mirrored bin Nc/2 = output R3/2 of butterfly 0
                const cf32 z = za[0][R3 / 2];
                pair(z, z, w_mid, (uint32_t)NC / 2, 0, 0u, 0, false);
            }
        }
    }
};

}  // namespace th
