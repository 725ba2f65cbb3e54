// stft_core.h — per-lane building blocks of the fused STFT → |X| → dB kernels.
//
// Everything here is written as small "phase" functions over (lane/thread id, registers, LDS
// pointer) so that the very same code is compiled (a) by hipcc into the gfx950 kernels in
// kernels_stft.hip and (b) by g++ into tests/emu/, which walks the lanes sequentially on the CPU
// to check index arithmetic without a GPU.  (b) is test scaffolding only — the product never runs
// it.
//
// Reference semantics implemented (paths relative to the reference checkout):
//   framing / reflect padding      src-tauri/src/core/spectrogram/stft.rs:16-149, core/utils.rs:111-138
//   window multiply + zero padding stft.rs:35-39,137-146
//   forward real FFT (realfft R2C) stft.rs:44-48  (restated: N/2-point complex FFT + split pass)
//   |X| = Complex::norm            spectrogram.rs:200
//   20*log10, 0 -> -inf            dynamics/decibel.rs:170-214
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define TH_HD __host__ __device__ __forceinline__
#else
#define TH_HD inline
#endif

// Pointers that reach a kernel inside a job table (loaded from memory) are "flat" to the compiler:
// it then emits flat_load / flat_store, which count on BOTH vmcnt and lgkmcnt and serialise with
// every LDS wait.  Kernels cast them to the global address space first (th::as_global).
#if defined(__HIP_DEVICE_COMPILE__)
#define TH_GLOBAL_AS __attribute__((address_space(1)))
#else
#define TH_GLOBAL_AS
#endif

namespace th {

// moment-form mel tables (build_mel_moments, mel_fuse.h <-> mel_moments_global, stft_wave.h): the group headers start at word 16
// (64-byte aligned) and are padded to whole batches of 4 groups — the kernel takes a batch's header as one s_load_dwordx8
constexpr uint32_t MEL_MOM_HDR0 = 16, MEL_MOM_BATCH = 4;
// words of an M block's masks (n taps, 2 words each; at least 128: the batch fetch reads that far behind a block's per-lane words)
TH_HD uint32_t mel_mom_mask_words(uint32_t n) { return 2u * n > 128u ? 2u * n : 128u; }
// The workgroup-per-frame kernels' LANE TABLE (build_mel_mom_lanes <-> mel_moments_range_lockstep): the same words [0] .. [8],
// no group header — group g's block sits at word MEL_LANE_BLK0 + MEL_LANE_STRIDE g, two planes of 64 x 4 words
constexpr uint32_t MEL_LANE_BLK0 = 16, MEL_LANE_STRIDE = 512;
constexpr uint32_t MEL_MOM_SPLIT8_BYTE = 16, MEL_MOM_SPLIT4_BYTE = 28;  // (mel_mom_splits, mel_fuse.h)

struct __attribute__((aligned(8))) cf32 {
    float re, im;
};

template <class T>
using gptr = T TH_GLOBAL_AS *;
template <class T>
TH_HD gptr<T> as_global(T *p) {
    return (gptr<T>)p;
}

// Exact power-of-two factor carried by the wave kernel's window table and by the generic kernel's magnitude: |X|^2
// would otherwise underflow below |X| ~ 1e-19 where the reference's hypot (spectrogram.rs:200) is finite (stft_wave.h).
constexpr float WAVE_PRESCALE = 4294967296.0f;          // 2^32
constexpr float WAVE_PRESCALE_DB = 192.65919722494797f;  // 20 log10(2^32)

TH_HD cf32 cadd(cf32 a, cf32 b) { return {a.re + b.re, a.im + b.im}; }
TH_HD cf32 csub(cf32 a, cf32 b) { return {a.re - b.re, a.im - b.im}; }
TH_HD cf32 cmul(cf32 a, cf32 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
TH_HD cf32 cmul_negi(cf32 a) { return {a.im, -a.re}; }  // a * (-i)
TH_HD cf32 cconj(cf32 a) { return {a.re, -a.im}; }

// Shapes shared by every STFT kernel (passed by value).
struct StftGeom {
    uint32_t hop, win, n_fft;
    uint32_t pad_left;   // (n_fft - win) / 2               stft.rs:36
    uint32_t nc;         // n_fft / 2  complex points of the packed FFT
    uint32_t log2_nc;
    uint32_t n_freq;     // n_fft / 2 + 1
    uint32_t height;     // columns of the output spec: n_freq (linear) or n_mel
    uint32_t n_mel;      // 0 = linear
    uint32_t frames_per_tile;
    uint32_t phased;     // wave kernel: 1 = frames are loaded from the 128-sample grid below their start (hop % 128 == 96)
    // n_fft = next_pow2(win) * f_overlap (spectrogram.rs:66-72) is a power of two times an ODD factor when f_overlap is not a
    // power of two (3, 5, 6, ...): nc = 2^log2_nc * (odd_m1 + 1).  0 for every power-of-two plan (round 5: the generic
    // kernel takes the odd factor as one more Stockham pass; the wave / block kernels are power-of-two only)
    uint32_t odd_m1;
};

// One frame range of one channel of the batch (device pointers).  A launch processes frames
// [f_begin, f_end) of the channel; its min/max goes to slot mm_index.
struct ChanJob {
    const float *wav;
    float *spec;
    uint32_t n_samples;
    uint32_t n_frames;
    uint32_t f_begin, f_end;
    uint32_t mm_index;
    uint32_t spec_pitch;  // floats per spec row (>= height; rows padded to 128 B keep every store line-aligned)
    uint32_t edge;        // wave kernel: 1 = boundary frames (reflect-padded fetch, one frame per chunk)
    uint32_t reserved;    // explicit tail padding (tables are compared bytewise before re-upload): always 0
};
static_assert(sizeof(ChanJob) == 48, "ChanJob must have no implicit padding");

// numpy-'reflect' index with periodic cycling for pads longer than N-1 (utils.rs:111-138;
// SURVEY.md Appendix A2).  n == 1 replicates the sample (the reference leaves that pad
// uninitialised, utils.rs:91,140).
TH_HD uint32_t reflect_index(int64_t i, uint32_t n) {
    if (i >= 0 && i < (int64_t)n) return (uint32_t)i;
    if (n == 1) return 0;
    const int64_t P = 2 * ((int64_t)n - 1);
    int64_t j = i % P;
    if (j < 0) j += P;
    return (uint32_t)(j < (int64_t)n ? j : P - j);
}

// Sample `i` (0 <= i < n_fft) of the zero-padded, windowed frame starting at signal position s0
// (= k*hop - win/2).  stft.rs:137-146
template <class WavPtr, class WinPtr>
TH_HD float frame_value(WavPtr wav, uint32_t n_samples, int64_t s0, uint32_t i, WinPtr window, const StftGeom &g) {
    if (i < g.pad_left || i >= g.pad_left + g.win) return 0.0f;
    const uint32_t wi = i - g.pad_left;
    return wav[reflect_index(s0 + (int64_t)wi, n_samples)] * window[wi];
}

// ---------------------------------------------------------------------------------------------
// Radix butterflies, forward transform (e^{-i...}), in registers.
// ---------------------------------------------------------------------------------------------
TH_HD void fft2(cf32 &a, cf32 &b) {
    const cf32 t = a;
    a = cadd(t, b);
    b = csub(t, b);
}

TH_HD void fft4(cf32 &a0, cf32 &a1, cf32 &a2, cf32 &a3) {
    const cf32 b0 = cadd(a0, a2), b1 = csub(a0, a2), b2 = cadd(a1, a3), b3 = cmul_negi(csub(a1, a3));
    a0 = cadd(b0, b2);
    a1 = cadd(b1, b3);
    a2 = csub(b0, b2);
    a3 = csub(b1, b3);
}

// ---------------------------------------------------------------------------------------------
// Generic workgroup kernel phases (any power-of-two n_fft; one frame at a time in LDS).
// Stockham autosort, decimation in time: pass with sub-transform size Ns and radix R maps
//   v_r = in[j + r*Nc/R] * W_{Ns*R}^{r*k},  k = j mod Ns;  out[(j-k)*R + k + r*Ns] = DFT_R(v)_r
// tw[i] = exp(-2*pi*i * i / n_fft), i in [0, n_fft).
// ---------------------------------------------------------------------------------------------
template <class WavPtr, class WinPtr>
TH_HD void gen_load(uint32_t tid, uint32_t nthr, const StftGeom &g, WavPtr wav, uint32_t n_samples, int64_t s0,
                    WinPtr window, cf32 *buf) {
    for (uint32_t n = tid; n < g.nc; n += nthr) {
        cf32 z;
        z.re = frame_value(wav, n_samples, s0, 2 * n, window, g);
        z.im = frame_value(wav, n_samples, s0, 2 * n + 1, window, g);
        buf[n] = z;
    }
}

TH_HD void gen_pass_r4(uint32_t tid, uint32_t nthr, const StftGeom &g, uint32_t Ns, const cf32 *tw,
                       const cf32 *in, cf32 *out) {
    const uint32_t q = g.nc >> 2;
    const uint32_t tw_step = g.n_fft / (Ns * 4);  // W_{Ns*4}^m = tw[m * tw_step]
    for (uint32_t j = tid; j < q; j += nthr) {
        const uint32_t k = j & (Ns - 1);
        cf32 v0 = in[j], v1 = in[j + q], v2 = in[j + 2 * q], v3 = in[j + 3 * q];
        if (Ns > 1) {
            v1 = cmul(v1, tw[k * tw_step]);
            v2 = cmul(v2, tw[2 * k * tw_step]);
            v3 = cmul(v3, tw[3 * k * tw_step]);
        }
        fft4(v0, v1, v2, v3);
        const uint32_t j0 = (j - k) * 4 + k;
        out[j0] = v0;
        out[j0 + Ns] = v1;
        out[j0 + 2 * Ns] = v2;
        out[j0 + 3 * Ns] = v3;
    }
}

TH_HD void gen_pass_r2(uint32_t tid, uint32_t nthr, const StftGeom &g, uint32_t Ns, const cf32 *tw,
                       const cf32 *in, cf32 *out) {
    const uint32_t h = g.nc >> 1;
    const uint32_t tw_step = g.n_fft / (Ns * 2);
    for (uint32_t j = tid; j < h; j += nthr) {
        const uint32_t k = j & (Ns - 1);
        cf32 v0 = in[j], v1 = in[j + h];
        if (Ns > 1) v1 = cmul(v1, tw[k * tw_step]);
        fft2(v0, v1);
        const uint32_t j0 = (j - k) * 2 + k;
        out[j0] = v0;
        out[j0 + Ns] = v1;
    }
}

// Pass with an ODD radix R (round 5: the odd part of Nc as ONE pass, R <= 63; Ns = the power-of-two part already done):
//   out[(j - k) R + k + c Ns] = sum_r in[j + r Nc / R] W_{Ns R}^{r k} W_R^{r c},   j < Nc / R, k = j mod Ns, c < R
// One output element per loop trip (consecutive threads = consecutive j: the reads in[j + r Nc / R] are contiguous per r);
// the two twiddles of a term are one table entry, W_{n_fft}^{r (k s1 + c s2)} with s1 = n_fft / (Ns R), s2 = n_fft / R.
TH_HD void gen_pass_odd(uint32_t tid, uint32_t nthr, const StftGeom &g, uint32_t Ns, uint32_t R, const cf32 *tw,
                        const cf32 *in, cf32 *out) {
    const uint32_t nb = g.nc / R, s1 = g.n_fft / (Ns * R), s2 = g.n_fft / R;
    for (uint32_t e = tid; e < g.nc; e += nthr) {
        const uint32_t c = e / nb, j = e - c * nb, k = j % Ns;
        const uint32_t step = (uint32_t)(((uint64_t)k * s1 + (uint64_t)c * s2) % g.n_fft);
        cf32 acc = in[j];  // r = 0
        uint32_t idx = 0;
        for (uint32_t r = 1; r < R; r++) {
            idx += step;
            if (idx >= g.n_fft) idx -= g.n_fft;
            const cf32 t = cmul(in[j + r * nb], tw[idx]);
            acc.re += t.re;
            acc.im += t.im;
        }
        out[(j - k) * R + k + c * Ns] = acc;
    }
}

// Split pass of the packed real FFT: from Z (Nc-point FFT of x[2n] + i*x[2n+1]) to the magnitudes
// of X[k] and X[Nc-k]:  E = (Z[k] + conj Z[Nc-k])/2, O = -i (Z[k] - conj Z[Nc-k])/2,
// T = W_{n_fft}^k O;  X[k] = E + T,  X[Nc-k] = conj(E - T).
TH_HD void split_pair(cf32 zk, cf32 zm, cf32 w, float &mag_k, float &mag_m) {
    const cf32 e = {0.5f * (zk.re + zm.re), 0.5f * (zk.im - zm.im)};
    const cf32 d = {0.5f * (zk.re - zm.re), 0.5f * (zk.im + zm.im)};
    const cf32 o = cmul_negi(d);
    const cf32 t = cmul(o, w);
    // (scaled by 2^32 before squaring, exact: re^2 + im^2 must not underflow where the reference's hypot does not)
    const cf32 xk = cadd(e, t), xm = csub(e, t);
    const float kr = xk.re * WAVE_PRESCALE, ki = xk.im * WAVE_PRESCALE, mr = xm.re * WAVE_PRESCALE, mi = xm.im * WAVE_PRESCALE;
    mag_k = __builtin_sqrtf(kr * kr + ki * ki) * (1.0f / WAVE_PRESCALE);  // correctly rounded (no fast-math)
    mag_m = __builtin_sqrtf(mr * mr + mi * mi) * (1.0f / WAVE_PRESCALE);
}

// dB_from_amp_inplace_default for one element — decibel.rs:179-202 (amin = 0, ref = 1):
// NaN or sign-negative -> NaN; x > 0 -> log10(x); +0 -> log10(0) - 0 = -inf; then * 20.
TH_HD float amp_to_dB(float x) {
    if (__builtin_isnan(x) || __builtin_signbit(x)) return __builtin_nanf("");
    return 20.0f * __builtin_log10f(x);
}

// Magnitudes of all n_freq bins of the frame whose packed FFT is in `z` (natural order) → mag[].
TH_HD void gen_split(uint32_t tid, uint32_t nthr, const StftGeom &g, const cf32 *tw, const cf32 *z, float *mag) {
    const uint32_t half = g.nc >> 1;
    for (uint32_t k = tid; k <= half; k += nthr) {
        const cf32 zk = z[k];
        const cf32 zm = z[k ? g.nc - k : 0u];  // (Nc is not a power of two when f_overlap is not)
        float mk, mm;
        split_pair(zk, zm, tw[k], mk, mm);
        mag[k] = mk;
        mag[g.nc - k] = mm;  // k = 0 writes mag[nc] (Nyquist); k = nc/2 writes the same bin twice
    }
}

// ---------------------------------------------------------------------------------------------
// Phases of the Bluestein kernel (stft_bluestein_kernel, kernels_stft.hip; the CPU emulator runs them with one "thread"):
// the packed Nc-point transform of a frame as a circular convolution of length M = 2^m >= 2 Nc - 1 with the chirp
// c[n] = e^{-i pi n^2 / Nc}, in double precision.  Tables: bluestein_tables (host_math.h).
// ---------------------------------------------------------------------------------------------
struct cf64 {
    double re, im;
};
TH_HD cf64 cmul64(cf64 a, cf64 b) { return {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re}; }
// a[n] = z[n] c[n] (z: the windowed samples as pairs, the product x * w in f32 as the reference forms it, stft.rs:137-146), 0 from Nc on
template <class WavPtr, class WinPtr>
TH_HD void bluestein_load(uint32_t tid, uint32_t nthr, const StftGeom &g, uint32_t M, WavPtr wav, uint32_t n_samples, int64_t s0, WinPtr window,
                          const cf64 *chirp, cf64 *buf) {
    for (uint32_t n = tid; n < M; n += nthr) {
        cf64 a = {0.0, 0.0};
        if (n < g.nc) {
            const cf64 z = {(double)frame_value(wav, n_samples, s0, 2 * n, window, g), (double)frame_value(wav, n_samples, s0, 2 * n + 1, window, g)};
            a = cmul64(z, chirp[n]);
        }
        buf[n] = a;
    }
}
// one radix-2 Stockham pass of the M-point transform (sub-transform size Ns; twm[k] = W_M^k, k < M / 2)
TH_HD void bluestein_pass(uint32_t tid, uint32_t nthr, uint32_t M, uint32_t Ns, const cf64 *twm, const cf64 *in, cf64 *out) {
    const uint32_t h = M >> 1, tw_step = h / Ns;  // W_{2 Ns}^k = W_M^(k M / (2 Ns))
    for (uint32_t j = tid; j < h; j += nthr) {
        const uint32_t k = j & (Ns - 1);
        const cf64 v0 = in[j], v1 = cmul64(in[j + h], twm[k * tw_step]);
        const uint32_t j0 = (j - k) * 2 + k;
        out[j0] = {v0.re + v1.re, v0.im + v1.im};
        out[j0 + Ns] = {v0.re - v1.re, v0.im - v1.im};
    }
}
// y <- conj(y FFT_M(b)): the inverse transform that follows is conj FFT conj
TH_HD void bluestein_product(uint32_t tid, uint32_t nthr, uint32_t M, const cf64 *bhat, cf64 *y) {
    for (uint32_t k = tid; k < M; k += nthr) {
        const cf64 c = cmul64(y[k], bhat[k]);
        y[k] = {c.re, -c.im};
    }
}
// Z[k] = c[k] conj(Y[k]) / M, k < Nc
TH_HD void bluestein_unchirp(uint32_t tid, uint32_t nthr, const StftGeom &g, uint32_t M, const cf64 *chirp, const cf64 *y, cf64 *z) {
    const double inv_m = 1.0 / (double)M;
    for (uint32_t k = tid; k < g.nc; k += nthr) z[k] = cmul64(cf64{y[k].re * inv_m, -y[k].im * inv_m}, chirp[k]);
}
// split pass of the packed real transform (split_pair) in f64, the magnitudes rounded once; tws[k] = W_{n_fft}^k, k <= Nc / 2
TH_HD void bluestein_split(uint32_t tid, uint32_t nthr, const StftGeom &g, const cf64 *tws, const cf64 *z, float *mag) {
    for (uint32_t k = tid; k <= g.nc / 2; k += nthr) {
        const cf64 zk = z[k], zm = z[k ? g.nc - k : 0u];
        const cf64 e = {0.5 * (zk.re + zm.re), 0.5 * (zk.im - zm.im)}, d = {0.5 * (zk.re - zm.re), 0.5 * (zk.im + zm.im)};
        const cf64 t = cmul64(cf64{d.im, -d.re}, tws[k]);
        const double kr = e.re + t.re, ki = e.im + t.im, mr = e.re - t.re, mi = e.im - t.im;
        mag[k] = (float)__builtin_sqrt(kr * kr + ki * ki);
        mag[g.nc - k] = (float)__builtin_sqrt(mr * mr + mi * mi);  // (k = 0 writes mag[nc], the Nyquist bin)
    }
}

}  // namespace th
