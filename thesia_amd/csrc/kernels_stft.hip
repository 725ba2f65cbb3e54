// kernels_stft.hip — fused STFT → |X| → [mel] → dB (+ per-channel min/max) kernels for gfx950.
//
// One launch processes a whole batch of independent channels (core/mod.rs:153-163).  A "tile" is
// `frames_per_tile` consecutive frames of one channel; blockIdx.x → tile through the per-channel
// prefix array tile_start[] (binary search, scalar loads).  Nothing but the f32 dB spec (which
// the reference retains, core/mod.rs:42) and 2 floats of min/max per channel is written to HBM:
// the windowed frames, the complex spectrum and the linear magnitudes never leave the CU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <type_traits>

#include "kernels.h"
#include "stft_core.h"
#include "stft_wave.h"
#include "stft_wave_multi.h"
#include "stft_block.h"

// Translation units (compile time: this file took 2 min 10 s of a 2 min 10 s build): compiled FOUR times — as itself
// (TH_STFT_PART undefined: every kernel but the one-frame wave kernels, every host-side helper, the dispatcher) and through
// kernels_stft_w1024.hip / _w2048.hip / _w4096.hip, which define TH_STFT_PART = 9 / 10 / 11 and compile the wave kernel's
// instantiations of that size and its launcher, nothing else.
#if defined(TH_STFT_PART)
#define TH_PART_MAIN 0
#else
#define TH_PART_MAIN 1
#endif

namespace th {

// ------------------------------------------------------------------------------------------
// float atomic min / max on plain f32 storage (sign-split integer trick; NaN never stored).
// Slot initialised to (+inf, -inf) by minmax_init_kernel.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_min_f32(float *addr, float v) {
    if (__builtin_isnan(v)) return;
    if (!__builtin_signbit(v)) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float *addr, float v) {
    if (__builtin_isnan(v)) return;
    if (!__builtin_signbit(v)) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

#if TH_PART_MAIN
__global__ void minmax_init_kernel(float *minmax, uint32_t n_chan) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (minmax != nullptr && i < n_chan) {
        minmax[2 * i] = __builtin_inff();
        minmax[2 * i + 1] = -__builtin_inff();
    }
}

#endif
// f32::min / f32::max semantics of the reference's scalar reductions (NaN-ignoring).
__device__ __forceinline__ float nmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float nmax(float a, float b) { return fmaxf(a, b); }

// three-input forms (one instruction; NaN-ignoring like v_min_f32 / v_max_f32: the inputs here are dB values, never signalling).
// -DTH_MINMAX3=1 folds two bins per v_min3_f32 / v_max3_f32 in the linear wave kernels (16 vector instructions fewer per frame of
// 699): measured inside bench.py's step on one card, alternating — 0.472 / 0.482 / 0.491 ms against 0.478 / 0.472 / 0.482 without
// (profiles/r05_ab_minmax3.txt): nothing, like every other cut of this kernel's instruction count.  Off.
#if !defined(TH_MINMAX3)
#define TH_MINMAX3 0
#endif
__device__ __forceinline__ float min3_f32(float a, float b, float c) {
    float r;
    asm("v_min3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ float max3_f32(float a, float b, float c) {
    float r;
    asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nmin(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nmax(v, __shfl_xor(v, o, 64));
    return v;
}

// blockIdx → (channel, first frame) via binary search over tile_start[0..n_chan]
__device__ __forceinline__ uint32_t find_chan(const uint32_t *__restrict__ tile_start, uint32_t n_chan,
                                              uint32_t tile) {
    uint32_t lo = 0, hi = n_chan;  // invariant: tile_start[lo] <= tile < tile_start[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (tile_start[mid] <= tile) lo = mid;
        else hi = mid;
    }
    return lo;
}

// ------------------------------------------------------------------------------------------
// Generic workgroup kernel: any power-of-two n_fft >= 2.  256 threads work on one frame at a time:
// ping-pong Stockham radix-4 (+ one radix-2 pass when log2(Nc) is odd), split pass, optional banded
// mel reduction, dB, coalesced row store.  Correctness-first fallback for sizes the wave kernel does
// not cover.  SCRATCH = false (n_fft <= 16384): the two frame buffers live in LDS, one workgroup per
// tile of frames.  SCRATCH = true (n_fft >= 32768: a 400 ms window at 48 kHz, 100 ms at 192 kHz —
// winMillisec has no upper bound in the UI, Control.tsx:96-107 / tracks.ts:205): the buffers are two
// n_fft/2-point regions of a global scratch area per workgroup (L2-resident: 256 KB at 32768) and a
// persistent grid walks the tiles; __syncthreads orders the workgroup's global writes and reads.
// ------------------------------------------------------------------------------------------
#if TH_PART_MAIN
constexpr int GEN_THREADS = 256;
// frames the LDS variant of the generic kernel runs side by side: 256 threads / the butterflies of a radix-4 pass (at least 16 threads a
// frame, at most a tile's 8 frames); 1 from n_fft 2048 on and wherever the buffers would not fit (the scratch variant)
__host__ __device__ inline uint32_t stft_generic_frames_par(const StftGeom &g) {
    const uint32_t per = g.nc / 4u < 16u ? 16u : g.nc / 4u;
    uint32_t fp = (uint32_t)GEN_THREADS / per;
    if (fp > 8u) fp = 8u;
    if (fp < 1u) fp = 1u;
    while (fp & (fp - 1u)) fp &= fp - 1u;  // (a power of two: nc / 4 need not be one when f_overlap is not)
    return fp;
}

template <bool SCRATCH>
__global__ __launch_bounds__(GEN_THREADS) void stft_generic_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ tile_start, uint32_t n_chan, uint32_t n_tiles,
    const float *__restrict__ window, const cf32 *__restrict__ tw, const float *__restrict__ mel_fb,
    const uint32_t *__restrict__ mel_lo, const uint32_t *__restrict__ mel_hi, float *__restrict__ minmax,
    cf32 *__restrict__ scratch) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    __shared__ float red[2 * (GEN_THREADS / 64)];

    const uint32_t tid = threadIdx.x;
    // Short transforms (round 6): a radix-4 pass of an Nc-point frame has Nc / 4 butterflies — 32 at n_fft 256, three quarters of the
    // workgroup idle through eight barriers per frame (408 M frames/s where the n_fft 512 wave kernel makes 1770).  The LDS variant
    // therefore runs FP = 256 / max(Nc / 4, 16) frames side by side, each on its own group of TPF threads and its own pair of
    // buffers, the barriers shared (stft_generic_frames_par; 8 frames at n_fft 256, a tile's worth).  Same arithmetic per element.
    const uint32_t FP = SCRATCH ? 1u : stft_generic_frames_par(g), TPF = GEN_THREADS / FP;
    const uint32_t sub = tid / TPF, ts = tid - sub * TPF;
    cf32 *bufA = SCRATCH ? scratch + (size_t)blockIdx.x * 2u * g.nc : reinterpret_cast<cf32 *>(smem_raw) + (size_t)sub * 2u * g.nc;
    cf32 *bufB = bufA + g.nc;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {  // (LDS variant: the grid is n_tiles, one trip)
    const uint32_t chan = find_chan(tile_start, n_chan, tile);
    const ChanJob job = jobs[chan];
    const uint32_t f0 = job.f_begin + (tile - tile_start[chan]) * g.frames_per_tile;
    const uint32_t f1 = min(f0 + g.frames_per_tile, job.f_end);

    float lmin = __builtin_inff(), lmax = -__builtin_inff();

    for (uint32_t fb = f0; fb < f1; fb += FP) {
        const uint32_t f = fb + sub;
        const bool act = f < f1;  // (a group without a frame in this round only keeps the barriers company)
        const int64_t s0 = (int64_t)f * g.hop - (int64_t)(g.win / 2);
        if (act) gen_load(ts, TPF, g, as_global(job.wav), job.n_samples, s0, as_global(window), bufA);
        __syncthreads();
        cf32 *in = bufA, *out = bufB;
        uint32_t Ns = 1;
        if (g.log2_nc & 1) {
            if (act) gen_pass_r2(ts, TPF, g, Ns, tw, in, out);
            __syncthreads();
            Ns = 2;
            cf32 *t = in; in = out; out = t;
        }
        const uint32_t odd = g.odd_m1 + 1u, n2 = g.nc / odd;  // nc = n2 * odd, n2 = 2^log2_nc
        for (; Ns < n2; Ns <<= 2) {
            if (act) gen_pass_r4(ts, TPF, g, Ns, tw, in, out);
            __syncthreads();
            cf32 *t = in; in = out; out = t;
        }
        if (odd > 1u) {  // f_overlap = 3, 5, 6, ... (spectrogram.rs:66-72): the odd factor as one more pass
            if (act) gen_pass_odd(ts, TPF, g, n2, odd, tw, in, out);
            __syncthreads();
            cf32 *t = in; in = out; out = t;
        }
        // `in` holds Z in natural order; magnitudes go to the other buffer (nc+1 floats fit in nc cf32)
        float *mag = reinterpret_cast<float *>(out);
        if (act) gen_split(ts, TPF, g, tw, in, mag);
        __syncthreads();
        const gptr<float> row = as_global(job.spec) + (size_t)f * job.spec_pitch;
        if (!act) {
        } else if (g.n_mel == 0) {
            for (uint32_t k = ts; k < g.n_freq; k += TPF) {
                const float d = amp_to_dB(mag[k]);
                row[k] = d;
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        } else {
            // linspec.dot(mel_fb) restricted to each filter's non-zero band, ascending f
            for (uint32_t m = ts; m < g.n_mel; m += TPF) {
                float acc = 0.0f;
                const uint32_t lo = mel_lo[m], hi = mel_hi[m];
                for (uint32_t k = lo; k < hi; k++) acc += mag[k] * mel_fb[(size_t)k * g.n_mel + m];
                const float d = amp_to_dB(acc);
                row[m] = d;
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        }
        __syncthreads();
    }

    if (minmax != nullptr) {
        lmin = wave_min(lmin);
        lmax = wave_max(lmax);
        if ((tid & 63) == 0) {
            red[2 * (tid >> 6)] = lmin;
            red[2 * (tid >> 6) + 1] = lmax;
        }
        __syncthreads();
        if (tid == 0) {
            float a = red[0], b = red[1];
            for (int w = 1; w < GEN_THREADS / 64; w++) {
                a = nmin(a, red[2 * w]);
                b = nmax(b, red[2 * w + 1]);
            }
            atomic_min_f32(&minmax[2 * job.mm_index], a);
            atomic_max_f32(&minmax[2 * job.mm_index + 1], b);
        }
        __syncthreads();  // (red[] is reused by the next tile of a persistent workgroup)
    }
  }
}

// ------------------------------------------------------------------------------------------
// Bluestein kernel (round 6): n_fft = 2^a * odd with an odd factor above 63 — f_overlap = 67, 71, 130, ...
// (spectrogram.rs:66-72: n_fft = next_pow2(win) * f_overlap for ANY integer f_overlap; the reference's realfft plans any
// length; no UI control offers such values).  The packed Nc-point transform of a frame (Nc = n_fft / 2, any integer) as a
// circular convolution of length M = 2^m >= 2 Nc - 1 with the chirp c[n] = e^{-i pi n^2 / Nc}:
//   Z[k] = c[k] * IFFT_M( FFT_M(z c) . FFT_M(b) )[k],   b[n] = b[M - n] = conj c[n]  (n < Nc), 0 elsewhere,
// in DOUBLE precision throughout (chirp products reach M |z|: in f32 the result would carry ~1e-5 of the frame maximum, above the
// 1e-3 dB bar of the parity tests; the card's f64 rate makes that free for a path no UI reaches): radix-2 Stockham passes over
// two M-point cf64 buffers per workgroup in global scratch, FFT_M(b) from the host, the inverse transform as conj FFT conj.
// The samples are windowed in f32 as the reference does (stft.rs:137-146); split pass and magnitudes in f64, rounded once.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(GEN_THREADS) void stft_bluestein_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ tile_start, uint32_t n_chan, uint32_t n_tiles,
    const float *__restrict__ window, const cf64 *__restrict__ chirp, const cf64 *__restrict__ bhat, const cf64 *__restrict__ twm,
    const cf64 *__restrict__ tws, const float *__restrict__ mel_fb, const uint32_t *__restrict__ mel_lo,
    const uint32_t *__restrict__ mel_hi, float *__restrict__ minmax, cf64 *__restrict__ scratch, uint32_t M) {
    cf64 *const bufA = scratch + (size_t)blockIdx.x * 2u * M, *const bufB = bufA + M;
    __shared__ float red[2 * (GEN_THREADS / 64)];
    const uint32_t tid = threadIdx.x;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint32_t chan = find_chan(tile_start, n_chan, tile);
        const ChanJob job = jobs[chan];
        const uint32_t f0 = job.f_begin + (tile - tile_start[chan]) * g.frames_per_tile;
        const uint32_t f1 = min(f0 + g.frames_per_tile, job.f_end);
        float lmin = __builtin_inff(), lmax = -__builtin_inff();
        for (uint32_t f = f0; f < f1; f++) {
            const int64_t s0 = (int64_t)f * g.hop - (int64_t)(g.win / 2);
            bluestein_load(tid, GEN_THREADS, g, M, as_global(job.wav), job.n_samples, s0, as_global(window), chirp, bufA);
            __syncthreads();
            cf64 *in = bufA, *out = bufB;
            for (int pass = 0; pass < 2; pass++) {  // FFT_M, then (behind the product with FFT_M(b)) the inverse as conj FFT conj
                for (uint32_t Ns = 1; Ns < M; Ns <<= 1) {
                    bluestein_pass(tid, GEN_THREADS, M, Ns, twm, in, out);
                    __syncthreads();
                    cf64 *t = in; in = out; out = t;
                }
                if (pass == 0) {
                    bluestein_product(tid, GEN_THREADS, M, bhat, in);
                    __syncthreads();
                }
            }
            bluestein_unchirp(tid, GEN_THREADS, g, M, chirp, in, out);  // Z in `out`
            __syncthreads();
            float *const mag = reinterpret_cast<float *>(in);  // (nc + 1 floats fit the nc cf64 of the transform's buffer)
            bluestein_split(tid, GEN_THREADS, g, tws, out, mag);
            __syncthreads();
            const gptr<float> row = as_global(job.spec) + (size_t)f * job.spec_pitch;
            if (g.n_mel == 0) {
                for (uint32_t k = tid; k < g.n_freq; k += GEN_THREADS) {
                    const float d = amp_to_dB(mag[k]);
                    row[k] = d;
                    lmin = nmin(lmin, d);
                    lmax = nmax(lmax, d);
                }
            } else {  // linspec.dot(mel_fb) restricted to each filter's non-zero band, ascending f (as the generic kernel)
                for (uint32_t m = tid; m < g.n_mel; m += GEN_THREADS) {
                    float acc = 0.0f;
                    const uint32_t lo = mel_lo[m], hi = mel_hi[m];
                    for (uint32_t k = lo; k < hi; k++) acc += mag[k] * mel_fb[(size_t)k * g.n_mel + m];
                    const float d = amp_to_dB(acc);
                    row[m] = d;
                    lmin = nmin(lmin, d);
                    lmax = nmax(lmax, d);
                }
            }
            __syncthreads();
        }
        if (minmax != nullptr) {
            lmin = wave_min(lmin);
            lmax = wave_max(lmax);
            if ((tid & 63) == 0) {
                red[2 * (tid >> 6)] = lmin;
                red[2 * (tid >> 6) + 1] = lmax;
            }
            __syncthreads();
            if (tid == 0) {
                float a = red[0], b = red[1];
                for (int w = 1; w < GEN_THREADS / 64; w++) {
                    a = nmin(a, red[2 * w]);
                    b = nmax(b, red[2 * w + 1]);
                }
                atomic_min_f32(&minmax[2 * job.mm_index], a);
                atomic_max_f32(&minmax[2 * job.mm_index + 1], b);
            }
            __syncthreads();
        }
    }
}

#endif  // TH_PART_MAIN (generic kernel)
// ------------------------------------------------------------------------------------------
// Wave kernel (the fast path): one 64-lane wave per frame, n_fft in {1024, 2048, 4096}, linear
// frequency scale.  See stft_wave.h for the FFT structure.
//
// Launch shape: a persistent grid of one workgroup per CU, WAVES waves each.  The workgroup shares
// read-only LDS tables (half-scaled zero-padded window, split-pass twiddles, pass-2 / pass-3
// twiddles) and gives every wave a private exchange slab:
//   LDS = 8 B * (2*Nc + T2_LEN + T3_LEN) + WAVES * 8 B * (Nc + Nc/R1)
// After the table fill there is no workgroup barrier: waves are independent.
//
// Work distribution: the interior frames of every channel are cut into CHUNKS of frames_per_tile
// consecutive frames; each wave pulls the next chunk index from a device-wide queue head (one
// returning atomicAdd per chunk) and walks its frames in order.  Walking consecutive frames is what
// makes the hop overlap free: with hop a multiple of 128 samples, frame f+1 is frame f shifted by
// S = hop/128 register slots, so a wave keeps its raw samples in registers, moves them S slots and
// loads only the S new slots (4 of 16 at hop = n_fft/4).  Every sample then reaches the CU once
// per chunk instead of once per overlapping frame (measured before: 2.4x read amplification at
// the fabric, because four waves requesting the same lines at the same time are not merged).
// The loads for frame f+1 are issued right after frame f's window multiply, a whole FFT ahead of
// their use.  HBM traffic per frame: 4*hop B read (+ (n_fft - hop)*4 B once per chunk) and
// 4*n_freq B written.
// ------------------------------------------------------------------------------------------
#define TH_SCHED_BARRIER() __builtin_amdgcn_sched_barrier(0)
// Row stores of the wave kernel: TH_STFT_NT bit 0 = non-temporal stores (the rows are written once and read by a later kernel,
// 1.5 GB behind), bit 1 = non-temporal sample loads.  A/B builds (profiles/r05_ab_stft_nt.txt); default 0.
#if !defined(TH_STFT_NT)
#define TH_STFT_NT 0
#endif
#if TH_STFT_NT & 1
#define TH_ROW_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define TH_ROW_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
// n_fft 2048: the packed-f32 pipeline (stft_pk.h, WaveFft<10>::*_pk) is template parameter PKV of wave_frame /
// stft_wave_kernel.  Round 4 built it as VERDICT r3 asked (v_pk_fma_f32 butterflies on pairs: 417 VALU instructions per frame
// instead of 681, 301 of them packed) and measured it against the scalar pipeline on one box, alternating, inside bench.py's
// step: 0.491 / 0.494 / 0.496 ms against 0.490 / 0.487 / 0.493 ms (profiles/r04_ab_packed.txt) — no gain: the launch does not
// follow its VALU instruction count.  The scalar pipeline stays the default; th_plan_set_kernel(plan, 9) runs the packed one
// on the headline shape (12 waves, hop = n_fft / 4, dB rows), parity-tested and reported next to the default by bench.py.
// -DTH_USE_PK=1 makes it the default of every n_fft 2048 launch shape (A/B builds).
#if !defined(TH_USE_PK)
#define TH_USE_PK 0
#endif

// (Development instrumentation — per-phase shader-clock totals, per-wave wall-clock stamps — is not part of this file: apply
// scripts/patches/instrumentation_phase_prof_wave_times.patch and build a variant, see scripts/phase_prof.py / wave_times.py.)

// Orders this wave's LDS writes before its later LDS reads (and vice versa) for the compiler; the
// hardware already executes one wave's DS instructions in order, so no instruction is needed.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A wave's position in the frame stream.  All members are wave-uniform (SGPRs).
struct FrameCursor {
    uint32_t f, f1, mm_index, spec_pitch, t;  // t: chunk index
    uint32_t chan, f_end;                     // job index and the end of its frame range (cursor_next)
    uint32_t n_samples, edge;
    gptr<const float> wav;
    gptr<float> spec;
    bool valid;
    bool fresh;  // first frame of a chunk: nothing of it is in registers yet
};

// Chunk t of the launch -> frame cursor.  chunk_tab[2 t] = job, chunk_tab[2 t + 1] = first frame (built by the host: one
// scalar load instead of a binary search over the jobs' first chunks — seven dependent loads from L2 per chunk start, which
// together with the queue atomic made a chunk start cost as much as 2.5 frames, scripts/wave_times.py).  t is wave-uniform.
__device__ __forceinline__ FrameCursor cursor_at(const StftGeom &g, const ChanJob *__restrict__ jobs,
                                                 const uint32_t *__restrict__ chunk_tab, uint32_t n_tiles, uint32_t t) {
    FrameCursor c{};
    c.valid = t < n_tiles;
    if (c.valid) {
        const uint32_t chan = chunk_tab[2 * (size_t)t];
        c.f = chunk_tab[2 * (size_t)t + 1];
        c.edge = jobs[chan].edge;
        c.chan = chan;
        c.f_end = jobs[chan].f_end;
        const uint32_t fpt = c.edge ? 1u : g.frames_per_tile;  // boundary frames: one per chunk (no register reuse)
        c.f1 = min(c.f + fpt, c.f_end);
        c.n_samples = jobs[chan].n_samples;
        c.mm_index = jobs[chan].mm_index;
        c.t = t;
        c.spec_pitch = jobs[chan].spec_pitch;
        c.wav = as_global(jobs[chan].wav);
        c.spec = as_global(jobs[chan].spec);
        c.fresh = true;
    }
    return c;
}

// The same for a chunk that usually continues the previous cursor's job (sweep schedule: consecutive 4-frame chunks): one
// 8-byte scalar load of the chunk table, and the job's fields only when the job changes.
__device__ __forceinline__ FrameCursor cursor_next(const StftGeom &g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab,
                                                   uint32_t n_tiles, uint32_t t, const FrameCursor &prev) {
    if (t >= n_tiles) return FrameCursor{};
    const uint2 e = *reinterpret_cast<const uint2 *>(chunk_tab + 2 * (size_t)t);
    if (!prev.valid || e.x != prev.chan) return cursor_at(g, jobs, chunk_tab, n_tiles, t);
    FrameCursor c = prev;
    c.f = e.y;
    c.f1 = min(c.f + (c.edge ? 1u : g.frames_per_tile), c.f_end);
    c.t = t;
    c.fresh = true;
    return c;
}

// The chunk schedule of a persistent wave (both wave kernels).  A wave's first chunk is its own index in the grid (no
// 3072-deep burst on the queue head at start); the rest come from the device-wide queue, one returning atomicAdd per chunk
// (the head counts from 0 behind the W static chunks, W = waves of the grid).  The pull is issued TH_PULL_AHEAD frames
// before the current chunk ends — late, so that the launch's last chunks still go to whoever runs ahead, but early enough
// that the atomic's round trip (served one at a time, ~8 ns each device-wide, and the waves of a launch tend to finish their
// chunks together) is over when the chunk is.
//     WaveSched s; TH_SCHED_INIT(s, wave index, frames ahead);
//     while (s.cur.valid) { fetch(s.cur); for (frames f) { TH_SCHED_PULL(s, f, lane); ... } TH_SCHED_ADVANCE(s, lane); }
#if !defined(TH_PULL_AHEAD)
#define TH_PULL_AHEAD 2u
#endif
struct WaveSched {
    FrameCursor cur;
    uint32_t pull_f, pulled, n_waves, ahead;
    bool use_queue, armed;
};
#define TH_SCHED_ARM(S) /* the frame at (or after) whose start the next chunk's index is pulled */          \
    ((S).pull_f = (S).cur.f1 - (S).cur.f > (S).ahead ? (S).cur.f1 - (S).ahead : (S).cur.f, (S).armed = (S).use_queue)
#define TH_SCHED_INIT(S, WAVE_INDEX, AHEAD) /* AHEAD >= the frame loop's step, so that the loop cannot step over pull_f */ \
    do {                                                                                                   \
        (S).ahead = (AHEAD);                                                                               \
        (S).n_waves = gridDim.x * WAVES;                                                                   \
        (S).use_queue = (S).n_waves < n_tiles; /* else: every chunk is some wave's first */                \
        (S).cur = cursor_at(g, jobs, chunk_tab, n_tiles, (WAVE_INDEX));                                    \
        (S).pulled = 0;                                                                                    \
        TH_SCHED_ARM(S);                                                                                   \
    } while (0)
#define TH_SCHED_PULL(S, F, LANE)                                                                          \
    do {                                                                                                   \
        if ((S).armed && (F) >= (S).pull_f) {                                                              \
            (S).armed = false;                                                                             \
            if ((LANE) == 0) (S).pulled = atomicAdd(queue_head, 1u);                                       \
        }                                                                                                  \
    } while (0)
#define TH_SCHED_ADVANCE(S, LANE)                                                                          \
    do {                                                                                                   \
        /* still armed: the frame loop never reached pull_f (a loop that advances by several frames may step over it; \
           without this pull the wave would walk the same chunk again, forever) */                         \
        if ((S).armed && (LANE) == 0) (S).pulled = atomicAdd(queue_head, 1u);                               \
        const uint32_t t_ = (S).use_queue ? __builtin_amdgcn_readfirstlane((S).pulled) + (S).n_waves : n_tiles; \
        (S).cur = cursor_at(g, jobs, chunk_tab, n_tiles, t_);                                              \
        TH_SCHED_ARM(S);                                                                                   \
    } while (0)

// frame element 0 sits at signal position f*hop - win/2 - pad_left; the host only hands interior
// frames to this kernel, so every windowed sample is inside the channel
__device__ __forceinline__ int64_t frame_e0(const FrameCursor &c, const StftGeom &g) {
    // (phased / dynamic mode: the load span starts on the 128-sample grid at or below the first window sample)
    const int64_t s0 = (int64_t)c.f * g.hop - (int64_t)(g.win / 2);
    return g.phased ? s0 - (s0 & 127) : s0 - (int64_t)g.pad_left;
}

__device__ __forceinline__ void flush_minmax(float *__restrict__ minmax, uint32_t slot, uint32_t lane, float lmin,
                                             float lmax) {
    const float a = wave_min(lmin), b = wave_max(lmax);
    if (lane == 0) {
        atomic_min_f32(&minmax[2 * slot], a);
        atomic_max_f32(&minmax[2 * slot + 1], b);
    }
}

constexpr uint32_t MEL_PRF_1024 = 512;  // pieces of the per-wave (r, f) buffer of the fused mel epilogue at n_fft = 1024

// One frame of the wave kernel (see stft_wave_kernel).  OFF = register rotation of x[] (logical slot m lives
// in physical x[(m + OFF) % P]); ROTATE = the caller instantiates one body per rotation instead of moving
// registers.  All state is passed as separate by-reference scalars / arrays: the body is always inlined and
// everything must stay in registers (a by-reference closure or struct here puts x[] into scratch memory).
// Advances cur to the next frame and returns true when that frame continues the same chunk.
// RES: which per-lane constant tables the caller keeps resident in registers instead of re-reading them from
// LDS every frame (bit 0 window, 1 pass-2 twiddles, 2 pass-3 twiddles, 3 split twiddles; 2 and 3 only on the
// mirror-local path).  Worth it when fewer waves per SIMD leave the VGPRs: LDS is a co-bottleneck of this kernel.
// PH >= 0: "phased" mode (stft_wave_kernel) — this frame sits at offset delta = (96 PH) mod 128 inside its n_fft-sample
// load span and wtab is the window table shifted by that much; the next frame's offset is (96 (PH + 1)) mod 128.
// PH == -2: "dynamic" mode — any hop in (128 SHIFT, 128 (SHIFT + 1)): the offset and the number of reused slots (SHIFT or
// SHIFT + 1) are computed per frame (wave-uniform), registers are moved instead of rotated, wtab = even table, the odd
// one (pairs shifted by one sample) NC + 64 entries behind it, each with 64 zero pairs in front.
// SWEEPF (sweep schedule, see stft_wave_kernel): this is the fourth and last frame of a chunk, and the NEXT chunk's first
// frame is requested in full — all P slots from (nwav, ne0) — where a frame otherwise requests the slots its successor in
// the chunk needs: after the window multiply every slot is dead in a chunk's last frame, so the next chunk starts without an
// exposed fetch and without an extra register.
// MELP (OUT == 3, the frame-pair mel epilogue): 1 = first frame of a pair — its amplitudes stay in registers (ampA, in the split
// pass's emit order) and nothing is written; 2 = second frame — its own amplitudes go to the second row of the slab, the first
// frame's from ampA to the first row, and one pass over the banded table (mel_banded_pair) forms both frames' mel rows.  A chunk
// that ends on a first frame is finished by wave_mel_flush.
// Slab layout of a frame pair: the second amplitude row RB floats behind the first (even; n_fft 2048: 1092 of the slab's 2176 floats,
// n_fft 1024: 544 of 1088), zeros behind each row: PAD_A (up to 64, below the second row) and PAD_B (to the end of the slab); the
// banded table may read min(PAD_A, PAD_B) floats past a row (the host checks the table's reach: stft_wave_mel_pair_applies)
template <int LOG2_NC>
struct MelPair {
    static constexpr int NC = 1 << LOG2_NC, SLAB_F = 2 * WaveFft<LOG2_NC>::SLAB_LEN;
    static constexpr int RB = LOG2_NC == 10 ? 1092 : SLAB_F / 2;
    static constexpr int PAD_A = RB - (NC + 1) < 64 ? RB - (NC + 1) : 64, PAD_B = SLAB_F - (RB + NC + 1) < 64 ? SLAB_F - (RB + NC + 1) : 64;
    static constexpr int PAD = PAD_A < PAD_B ? PAD_A : PAD_B;
    static_assert(RB % 2 == 0 && PAD_A > 0 && PAD_B > 0, "two rows and their zeros fit the slab");
};
constexpr int MEL_PAIR_RB = MelPair<10>::RB;
#if !TH_PART_MAIN  // the one-frame wave kernel: the per-size translation units
template <int LOG2_NC, int SHIFT, int OUT, bool ROTATE, int OFF, int RES, int PH = -1, bool PKV = false, bool SWEEPF = false, int MELP = 0>
__device__ __forceinline__ void wave_frame(
    const StftGeom &g, const cf32 *wtab, const cf32 *stw, const cf32 *t2, const cf32 *t3, cf32 *slab, uint32_t lane_wave,
    uint32_t f, uint32_t f1, gptr<const float> wav, uint32_t n_samples, gptr<float> spec, uint32_t spec_pitch, cf32 (&x)[WaveFft<LOG2_NC>::P],
    const cf32 (&rw)[(RES & 1) ? WaveFft<LOG2_NC>::P : 1], const cf32 (&rw2)[(RES & 2) ? WaveFft<LOG2_NC>::NT2 : 1],
    const cf32 (&rwa)[(RES & 4) ? WaveFft<LOG2_NC>::NQ : 1][WaveFft<LOG2_NC>::NT3],
    const cf32 (&rwb)[(RES & 4) ? WaveFft<LOG2_NC>::NQ : 1][WaveFft<LOG2_NC>::NT3],
    const cf32 (&rws)[(RES & 8) ? WaveFft<LOG2_NC>::NQ : 1][WaveFft<LOG2_NC>::R3], cf32 rw_mid, float &lmin, float &lmax,
    const uint32_t *meltab, cf32 *mel_prf, const WaveOut &wo, float (&ampA)[OUT == 3 ? WaveFft<LOG2_NC>::N_EMIT : 1], bool nxt_full = false,
    gptr<const float> nwav = nullptr, int64_t ne0 = 0) {
    constexpr bool AMP = OUT == 1, MELF = OUT == 2 || OUT == 3 || OUT == 4;  // (4: the moment form of the one-frame epilogue where no LDS table form exists, n_fft 1024 / 2048)
    static_assert((OUT == 3) == (MELP != 0), "frame pairs: OUT = 3 with MELP = 1 (first frame) or 2 (second)");
    static_assert(OUT != 3 || ((LOG2_NC == 10 || LOG2_NC == 9) && !PKV), "frame pairs: n_fft 1024 / 2048, scalar pipeline");
    using MP = MelPair<(LOG2_NC == 9 || LOG2_NC == 10) ? LOG2_NC : 10>;
    using W = WaveFft<LOG2_NC>;
    constexpr int P = W::P, NC = W::NC;
    // Per-frame opaque copy of the lane id.  Everything below addresses LDS and the output row as
    // "f(lane) + immediate"; left loop-invariant, LICM hoists ~40 such addresses out of the frame loop
    // and the register allocator spills them.  Recomputing the handful of bases per frame is cheaper.
    uint32_t lane = lane_wave;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(lane));
#endif
    lane &= 63u;
    const uint32_t col = W::lane_col(lane);  // the lane's column of the frame: complex points col + 64 m (stft_wave.h)
    // LDS reads return in order and a read issued next to its use exposes the whole LDS latency, so all
    // table reads are issued ahead of their use (lds_ld keeps program order): pass-2 twiddles before the
    // pass-1 arithmetic, pass-3 and split twiddles together with the reads of exchange 2.
    cf32 z[P];
    cf32 w2[W::NT2];
    constexpr bool PKP = W::PK && PKV;
    v2f zp[PKP ? P : 1];  // the windowed frame as (re, im) pairs (packed pipeline)
    // PH == -3: dynamic mode with even offsets only and NO zero pairs in front of the (single) table — the table's own tail is
    // zeros (n_fft - win >= 128 samples), so the one read that would fall below the table wraps around into it
    constexpr bool DYN = PH == -2 || PH == -3, DYNW = PH == -3;
    if constexpr (DYNW) {
        const uint32_t h = ((uint32_t)((int64_t)f * g.hop - (int64_t)(g.win / 2)) & 127u) >> 1;  // (the offset is even) pairs above the grid
        {
            const cf32 w = lds_ld(&wtab[(col - h) & (uint32_t)(NC - 1)]);
            z[0] = {x[0].re * w.re, x[0].im * w.im};
        }
        const cf32 *const wb = wtab + col - h;  // (slots m >= 1 start at pair 64 m + col - h >= 1)
#pragma unroll
        for (int m = 1; m < P; m++) {
            const cf32 w = lds_ld(&wb[64u * m]);
            z[m] = {x[m].re * w.re, x[m].im * w.im};
        }
    } else if constexpr (DYN) {
        const uint32_t d = (uint32_t)((int64_t)f * g.hop - (int64_t)(g.win / 2)) & 127u;  // first window sample above the grid
        wave_window_rot<P, 0>(col, z, x, wtab + 64 + ((d & 1u) ? NC + 64 : 0) - ((d + 1u) >> 1));
    } else if constexpr ((RES & 1) != 0 && PKP) {
#pragma unroll
        for (int m = 0; m < P; m++) {
            const cf32 v = x[(m + OFF) % P];
            zp[m] = mk2(v.re, v.im) * mk2(rw[m].re, rw[m].im);  // one v_pk_mul_f32: (x[2n] w[2n], x[2n+1] w[2n+1])
        }
    } else if constexpr (RES & 1) {
#pragma unroll
        for (int m = 0; m < P; m++) {
            const cf32 v = x[(m + OFF) % P];
            z[m] = {v.re * rw[m].re, v.im * rw[m].im};
        }
    } else {
        wave_window_rot<P, OFF>((W::PLANES32 || (W::PLANES8 && PH == -1)) ? lane : col, z, x, wtab);  // (4096, 1024: table stored in lane order, see the kernel)
    }
    if constexpr (PKP && (DYN || !(RES & 1))) {
#pragma unroll
        for (int m = 0; m < P; m++) zp[m % (PKP ? P : 1)] = mk2(z[m].re, z[m].im);
    }
    // Request the next frame of the chunk now: its samples land while this frame is transformed.  The fetch
    // is unconditional (branch-free register flow: no copies of x[]); on the last frame of a chunk it simply
    // re-reads this frame's span, which is in bounds, and the result is never used.
    {
        const uint32_t fn = f + 1 < f1 ? f + 1 : f;
        // (clamped into the channel: the one-frame chunks of boundary frames prefetch "themselves", and that span is
        // partly outside; a no-op for interior frames)
        const int64_t s_n = (int64_t)fn * g.hop - (int64_t)(g.win / 2);  // first window sample of the next frame
        const int64_t lead_n = DYN ? (s_n & 127) : PH >= 0 ? (int64_t)((96 * (PH + 1)) & 127) : (int64_t)g.pad_left;
        int64_t e0n = s_n - lead_n;
        bool one_more = false;  // dynamic mode: the next frame's load span starts SHIFT + 1 slots further, not SHIFT
        if constexpr (DYN) {
            const int64_t s_c = (int64_t)f * g.hop - (int64_t)(g.win / 2);
            one_more = e0n - (s_c - (s_c & 127)) > 128 * SHIFT;
        }
        const int64_t e0_max = (int64_t)n_samples - (int64_t)g.n_fft;
        e0n = e0n < 0 ? 0 : (e0n > e0_max ? e0_max : e0n);
        if constexpr (DYN) {
            if (one_more) {  // wave-uniform
#pragma unroll
                for (int m = 0; m + SHIFT + 1 < P; m++) x[m] = x[m + SHIFT + 1];
                wave_fetch<P, P - SHIFT - 1>(col, x, wav, e0n);
            } else {
#pragma unroll
                for (int m = 0; m + SHIFT < P; m++) x[m] = x[m + SHIFT];
                wave_fetch<P, P - SHIFT>(col, x, wav, e0n);
            }
        } else if constexpr (SHIFT == 0) {
            wave_fetch<P, 0>(col, x, wav, e0n);
        } else if constexpr (ROTATE && SWEEPF) {
            // (unconditional: a wave-uniform branch between the two fetches made the register allocator spill loaded slots
            // behind a vmcnt(0) in every frame; the caller passes this frame's own span when there is nothing to prefetch)
            (void)nxt_full;
            wave_fetch<P, 0>(col, x, nwav, ne0);
        } else if constexpr (ROTATE) {
            wave_fetch_rot<P, SHIFT, OFF>(col, x, wav, e0n);
        } else {
#pragma unroll
            for (int m = 0; m + SHIFT < P; m++) x[m] = x[m + SHIFT];
            wave_fetch<P, P - SHIFT>(col, x, wav, e0n);
        }
    }
    TH_SCHED_BARRIER();
    const gptr<float> row = spec + (size_t)f * spec_pitch;
    if constexpr (PKP) {
        // packed-f32 pipeline (stft_pk.h; WaveFft::pass1_pk .. split_paired_pk): the same phases, exchanges and waits
        v2f w2p[W::NT2];
        if constexpr (!(RES & 2)) {
            W::load_t2_pk(lane, w2p, t2);  // lands during the pass-1 arithmetic
        } else {
#pragma unroll
            for (int r = 0; r < W::NT2; r++) w2p[r] = mk2(rw2[r].re, rw2[r].im);
        }
        W::pass1_pk(lane, zp, slab);
        wave_lds_sync();
        TH_SCHED_BARRIER();
        v2f zr[8], zi[8];
        W::read1_pk(lane, zr, zi, slab);
        wave_lds_sync();
        W::pass2_pk(lane, zr, zi, w2p, slab);
        wave_lds_sync();
        TH_SCHED_BARRIER();
        v2f wa[W::NQ][W::NT3], wb[W::NQ][W::NT3];
        const typename W::PairBase pbs = W::pair_base(lane);
        if constexpr (!(RES & 4)) {
            W::load_t3_paired_pk(pbs, wa, wb, t3);  // queued behind the exchange writes, ahead of the exchange reads
        } else {
#pragma unroll
            for (int q = 0; q < W::NQ; q++)
#pragma unroll
                for (int r = 0; r < W::NT3; r++) {
                    wa[q][r] = mk2(rwa[q][r].re, rwa[q][r].im);
                    wb[q][r] = mk2(rwb[q][r].re, rwb[q][r].im);
                }
        }
        typename W::PkPairs za[W::NQ], zb[W::NQ];
        W::read2_paired_pk(pbs, za, zb, slab);
        wave_lds_sync();  // slab is free again: the next frame's pass 1 may overwrite it
        v2f wsr[W::NQ][2], wsi[W::NQ][2];
        if constexpr (!(RES & 8)) {
            W::load_stw_paired_pk(lane, wsr, wsi, stw);  // (stw = the pair-ordered table, WaveFft::fill_stwp)
        } else {
#pragma unroll
            for (int q = 0; q < W::NQ; q++)
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    wsr[q][h] = mk2(rws[q][h].re, rws[q][h + 2].re);
                    wsi[q][h] = mk2(rws[q][h].im, rws[q][h + 2].im);
                }
        }
        TH_SCHED_BARRIER();
        W::pass3_paired_pk(za, zb, wa, wb);
        TH_SCHED_BARRIER();
        float *const slab_f = reinterpret_cast<float *>(slab);
        auto row_at = [&](uint32_t kb, int kc) -> gptr<float> {
            return (gptr<float>)((gptr<char>)row + (uint64_t)(kb << 2) + (int64_t)(4 * kc));
        };
        auto emit = [&](uint32_t kb, int kc, float p) {
            if constexpr (MELF) {
                slab_f[kb + (uint32_t)kc] = power_to_amp_scaled(p);
            } else if constexpr (AMP) {
                TH_ROW_STORE(row_at(kb, kc), power_to_amp(p));
            } else {
                const float d = power_to_dB(p);
                TH_ROW_STORE(row_at(kb, kc), d);
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        };
        W::split_paired_pk(lane, za, zb, wsr, wsi, rw_mid, emit);
    } else {
    if constexpr (!(RES & 2)) W::load_t2(lane, w2, t2);  // lands during the pass-1 arithmetic
    W::pass1(lane, z, slab);
    wave_lds_sync();
    TH_SCHED_BARRIER();
    W::read1(lane, z, slab);
    wave_lds_sync();

    if constexpr (W::PAIRED) {
        // mirror-local last pass: every Z[k] / Z[Nc-k] pair ends up in one lane's registers
        if constexpr (RES & 2) W::pass2_w(lane, z, rw2, slab);
        else W::pass2_w(lane, z, w2, slab);
        wave_lds_sync();
        TH_SCHED_BARRIER();
        cf32 wa[W::NQ][W::NT3], wb[W::NQ][W::NT3];
        const typename W::PairBase pbs = W::pair_base(lane);  // the lane's butterfly pairs as base + immediate (once per frame)
        if constexpr (!(RES & 4)) W::load_t3_paired(pbs, wa, wb, t3);  // queued behind the exchange writes, ahead of the exchange reads
        cf32 za[W::NQ][W::R3], zb[W::NQ][W::R3];
        W::read2_paired(lane, pbs, za, zb, slab);
        wave_lds_sync();  // slab is free again: the next frame's pass 1 may overwrite it
        constexpr bool PRELOAD_STW = W::NQ * W::R3 <= 8;  // 16 VGPRs at n_fft = 2048; too many at 4096
        cf32 ws[W::NQ][W::R3];
        if constexpr (PRELOAD_STW && !(RES & 8)) W::load_stw_paired(lane, ws, stw);
        TH_SCHED_BARRIER();
        if constexpr (RES & 4) W::pass3_paired_w(za, zb, rwa, rwb);
        else W::pass3_paired_w(za, zb, wa, wb);
        TH_SCHED_BARRIER();
        if constexpr (!PRELOAD_STW && !(RES & 8)) W::load_stw_paired(lane, ws, stw);
        float *const slab_f = reinterpret_cast<float *>(slab);
        // bin kb + kc: per-lane base (opaque, split_base) + compile-time constant.  The global address is formed as
        // "row (SGPRs) + zext(4 kb) + 4 kc" in 64 bits, which is exactly global_store's saddr + voffset + immediate.
        auto row_at = [&](uint32_t kb, int kc) -> gptr<float> {
            return (gptr<float>)((gptr<char>)row + (uint64_t)(kb << 2) + (int64_t)(4 * kc));
        };
        int amp_i = 0;  // (a constant at every call once the split pass is unrolled)
        int mm_i = 0;
        float mm_pend = __builtin_nanf("");  // (ignored by min / max until the first bin replaces it)
        auto emit = [&](uint32_t kb, int kc, float p) {
            if constexpr (MELP == 1) {
                ampA[amp_i++ % (OUT == 3 ? W::N_EMIT : 1)] = power_to_amp_scaled(p);  // first frame of a pair: kept for the second one's epilogue
            } else if constexpr (MELF) {
                slab_f[(MELP == 2 ? (uint32_t)MP::RB : 0u) + kb + (uint32_t)kc] = power_to_amp_scaled(p);  // the slab is free after the second exchange: amplitude row for mel_pieces
            } else if constexpr (AMP) {
                TH_ROW_STORE(row_at(kb, kc), power_to_amp(p));
            } else {
                const float d = power_to_dB(p);
                TH_ROW_STORE(row_at(kb, kc), d);
#if TH_MINMAX3
                // two bins per v_min3_f32 / v_max3_f32 (round 5: the running min / max were 34 of the frame's 699 vector instructions);
                // mm_i is a constant at every call once the split pass is unrolled
                if ((mm_i++ & 1) == 0) {
                    mm_pend = d;
                } else {
                    lmin = min3_f32(lmin, mm_pend, d);
                    lmax = max3_f32(lmax, mm_pend, d);
                }
#else
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
#endif
            }
        };
        if constexpr (RES & 8) W::split_paired_w(lane, za, zb, rws, rw_mid, emit);
        else W::split_paired_w(lane, za, zb, ws, stw[NC / 2], emit);
#if TH_MINMAX3
        if constexpr (!MELF && !AMP) {  // lane 0's seventeenth bin (elsewhere a value that has been folded already: idempotent)
            lmin = nmin(lmin, mm_pend);
            lmax = nmax(lmax, mm_pend);
        }
#endif
    } else {
        // n_fft = 1024: one last-pass butterfly per lane; the halves of the wave swap registers for the split pass (no third
        // exchange, stft_wave.h)
        static_assert(W::SWAP8, "plans without mirror-local pairs use the half-wave swap");
        if constexpr (RES & 2) W::pass2_w(lane, z, rw2, slab);
        else W::pass2_w(lane, z, w2, slab);
        wave_lds_sync();
        TH_SCHED_BARRIER();
        const typename W::Swap8Lane sl = W::swap8_lane(lane);
        cf32 w3[4], z256;
        W::load_t3_sw(sl, w3, t3);
        W::read2_sw(sl, z, slab);
        wave_lds_sync();  // the slab is free: the next frame's pass 1 (or the amplitude row of the mel epilogue) rewrites it
        TH_SCHED_BARRIER();
        W::pass3_sw(sl, z, w3, z256);
        W::mirror_swap(sl, z);
        float *const slab_f = reinterpret_cast<float *>(slab);
        int amp_i = 0;  // (a constant at every call once split_sw is unrolled)
        auto emit = [&](uint32_t kb, int kc, float p) {  // bin kb + kc: per-lane base + compile-time constant
            if constexpr (MELP == 1) {
                ampA[amp_i++ % (OUT == 3 ? W::N_EMIT : 1)] = power_to_amp_scaled(p);  // first frame of a pair (see the mirror-local path)
            } else if constexpr (MELF) {
                slab_f[(MELP == 2 ? (uint32_t)MP::RB : 0u) + kb + (uint32_t)kc] = power_to_amp_scaled(p);
            } else {
                const float d = AMP ? power_to_amp(p) : power_to_dB(p);
                *(gptr<float>)((gptr<char>)row + (uint64_t)(kb << 2) + (int64_t)(4 * kc)) = d;
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        };
        W::split_sw(sl, z, z256, stw, emit);
    }
    }  // (scalar pipeline)
    if constexpr (MELP == 2) {
        // frame pair: this frame's amplitudes are in the slab's second row; lay the first frame's (f - 1) out as the first row,
        // zeros behind both rows, then one pass over the banded table for both (mel_banded_pair, stft_wave.h)
        float *const slab_f = reinterpret_cast<float *>(slab);
        {
            int k = 0;
            W::split_enumerate(lane, [&](uint32_t kb, int kc) { slab_f[kb + (uint32_t)kc] = ampA[k++ % W::N_EMIT]; });
        }
        if (MP::PAD_A >= 64 || lane < (uint32_t)MP::PAD_A) slab_f[NC + 1 + lane] = 0.0f;  // below the second row
        if (MP::PAD_B >= 64 || lane < (uint32_t)MP::PAD_B) slab_f[MP::RB + NC + 1 + lane] = 0.0f;
        wave_lds_sync();
        const gptr<float> row_a = row - spec_pitch;
        float mn = __builtin_inff(), mx = -__builtin_inff();  // of the pair's rows (values fresh from an FMA: no canonicalising per group)
        // (lanes past the last mel repeat the last filter — build_mel_band — so every lane's value may enter min / max; only the
        // store is masked)
        auto emit_a = [&](uint32_t m, float v) {
            const float d = amp_to_dB_fast(v);
            mn = nmin(mn, d);
            mx = nmax(mx, d);
            if (m < wo.n_mel) row_a[m] = d;
        };
        auto emit_b = [&](uint32_t m, float v) {
            const float d = amp_to_dB_fast(v);
            mn = nmin(mn, d);
            mx = nmax(mx, d);
            if (m < wo.n_mel) row[m] = d;
        };
#if defined(TH_MELF_ABL) && (TH_MELF_ABL & 1)
        if (wo.n_mel == 0x7fffffffu)
#endif
        mel_banded_pair<MP::RB, LOG2_NC != 9>(lane, slab_f, meltab, wo.mel_groups, wo.band_off, wo.band_n, emit_a, emit_b);
        lmin = nmin(lmin, mn);
        lmax = nmax(lmax, mx);
        wave_lds_sync();  // the next frame's pass 1 rewrites the slab
        const uint32_t pad = spec_pitch - wo.n_mel;  // complete both rows' last 128-byte line (see below)
        if (lane - 1u < ((pad < 32u && spec_pitch % 32u == 0) ? pad : 0u)) {
            row_a[wo.n_mel - 1u + lane] = 0.0f;
            row[wo.n_mel - 1u + lane] = 0.0f;
        }
    } else if constexpr (MELF && MELP == 0) {
        // fused mel filterbank (stft_wave.h / mel_fuse.h): the frame's amplitudes sit in the wave's slab; pieces of 4 bins
        // -> (r, f) partial sums -> one mel per lane and group; the (r, f) buffer sits behind the amplitude row
        float *const slab_f = reinterpret_cast<float *>(slab);
        // the (r, f) buffer: behind the amplitude row in the slab (n_fft 2048: room for 512 pieces), or the wave's own
        // region behind the mel table (n_fft 1024: the slab would only hold 256 pieces, the default mel counts need ~400)
        auto emit_mel = [&](uint32_t m, float v) {
            if (m < wo.n_mel) {
                const float d = amp_to_dB_fast(v);
                row[m] = d;
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        };
#if defined(TH_MELF_ABL) && (TH_MELF_ABL & 1)  // ablation build: no filterbank sums (rows unwritten) — what the epilogue costs
        if (wo.n_mel == 0x7fffffffu)
#endif
        if constexpr (LOG2_NC == 11 || OUT == 4) {
            // no LDS for a table beside eight slabs (n_fft 1024 / 2048, OUT = 4: more mels than an LDS table holds): the moment form (lane = segment, two additions per bin; per-lane words and the
            // taps' lane masks from global memory — mel_moments_global, stft_wave.h; build_mel_moments, mel_fuse.h)
            wave_lds_sync();
            mel_moments_global(lane, slab_f, as_global(wo.mel_tab), wo.mel_groups, emit_mel);
        } else if (wo.mel_slots == 0) {  // wave-uniform: banded sums, lane = mel (mel_banded, stft_wave.h; table: build_mel_band)
            // the filters of a group reach up to its widest one's width past their own end: zeros behind the row
            static_assert(2 * (int)W::SLAB_LEN >= NC + 1 + 128, "room for the zeros behind the amplitude row");
            slab_f[NC + 1 + lane] = 0.0f;
            slab_f[NC + 65 + lane] = 0.0f;
            wave_lds_sync();
            mel_banded<TH_MEL_BAND_PAIRED != 0>(lane, slab_f, meltab, wo.mel_groups, wo.band_off, wo.band_n, emit_mel);
        } else {
            cf32 *const prf = mel_prf != nullptr ? mel_prf : slab + (NC + 2) / 2;
            const MelFuseTab mt = mel_fuse_view(meltab, wo.mel_slots, wo.mel_groups);
            wave_lds_sync();
            mel_pieces(lane, slab_f, prf, mt);
            wave_lds_sync();
            mel_gather(lane, prf, mt, emit_mel);
        }
        wave_lds_sync();  // the next frame's pass 1 rewrites the slab
    }
    // Rows at the library's padded pitch (th_pitch_f32): bin Nc would be the only dword written in its 128-byte line, and
    // a partially written line costs HBM a read-modify-write (scripts/ubench/row_stores.hip: 3.9 -> 5.4 TB/s for this row
    // shape).  The padding is ours, so complete the line with zeros.
    if constexpr (MELP == 0) {
        const uint32_t height = MELF ? wo.n_mel : (uint32_t)(NC + 1);
        const uint32_t pad = spec_pitch - height;
        if (lane - 1u < ((pad < 32u && spec_pitch % 32u == 0) ? pad : 0u)) row[height - 1u + lane] = 0.0f;
    }
    TH_SCHED_BARRIER();
}

// Frame pairs (OUT == 3): a chunk ended on the first frame of a pair — frame f's amplitudes are in ampA.  Lay them out as a row
// and run the one-frame banded sums (what wave_frame<.., OUT = 2> does behind its split pass).
template <int LOG2_NC>
__device__ __forceinline__ void wave_mel_flush(cf32 *slab, uint32_t lane_wave, uint32_t f, gptr<float> spec, uint32_t spec_pitch,
                                               const float (&ampA)[WaveFft<LOG2_NC>::N_EMIT], float &lmin, float &lmax, const uint32_t *meltab,
                                               const WaveOut &wo) {
    using W = WaveFft<LOG2_NC>;
    constexpr int NC = W::NC;
    uint32_t lane = lane_wave;
    asm volatile("" : "+v"(lane));
    lane &= 63u;
    float *const slab_f = reinterpret_cast<float *>(slab);
    const gptr<float> row = spec + (size_t)f * spec_pitch;
    {
        int k = 0;
        W::split_enumerate(lane, [&](uint32_t kb, int kc) { slab_f[kb + (uint32_t)kc] = ampA[k++ % W::N_EMIT]; });
    }
    slab_f[NC + 1 + lane] = 0.0f;
    slab_f[NC + 65 + lane] = 0.0f;
    wave_lds_sync();
    auto emit_mel = [&](uint32_t m, float v) {
        if (m < wo.n_mel) {
            const float d = amp_to_dB_fast(v);
            row[m] = d;
            lmin = nmin(lmin, d);
            lmax = nmax(lmax, d);
        }
    };
    mel_banded<true>(lane, slab_f, meltab, wo.mel_groups, wo.band_off, wo.band_n, emit_mel);
    wave_lds_sync();
    const uint32_t pad = spec_pitch - wo.n_mel;
    if (lane - 1u < ((pad < 32u && spec_pitch % 32u == 0) ? pad : 0u)) row[wo.n_mel - 1u + lane] = 0.0f;
}

// SHIFT = hop/128 register slots reused between consecutive frames (0 = no reuse: hop not a
// multiple of 128 samples or hop >= n_fft)
// AMP: store the linear amplitude |X| instead of dB and skip min/max (first half of the mel path;
// mel_mfma_kernel then applies the filterbank).
// SWEEP (round 4; rotation mode with NROT = 4, i.e. hop = n_fft / 4): the "sweep" chunk schedule for large batches.  Chunks
// are 4 frames (one rotation cycle) and are dealt out IN ORDER, so that everything in flight chip-wide is one compact window
// moving linearly through the input and the output — what the DRAM likes (r03 stream_shapes: 5.9-6.4 TB/s against 4.9-5.1 for
// 3072 independent 126 KB streams; scripts/ubench/stft_skeleton.hip mode 8 with this very schedule: 3-10 % off the memory
// skeleton, 8-10 % with the kernel's FMA work) — without giving up the register reuse or the dynamic balance, and without a
// workgroup barrier:
//   * a workgroup takes BLOCKS of SWEEP_BLK = 12 consecutive chunks from the device-wide in-order queue (one returning
//     atomicAdd per block, 7.5 k per launch of the bench workload where per-chunk pulls would be 90 k at ~8 ns each), one block
//     ahead of its use; its first block is its own index;
//   * its waves draw chunks of the current block from a ticket counter in LDS (the three waves of a SIMD run at different
//     speeds: dynamic inside the workgroup), so the 12 waves of a CU write 12 neighbouring 16 KB pieces at any time;
//   * the first frame of a wave's next chunk is requested in the last frame of its current one (wave_frame, SWEEPF), its
//     cursor having been drawn and looked up a frame earlier.
// The (job, first frame) chunk table, the per-chunk (min, max) slots and wave_post_kernel's fold are the ones of the default
// schedule; queue_head counts blocks here.
constexpr uint32_t SWEEP_BLK = 12, SWEEP_RING = 8;
// (the kernel's body as an inlined device function: two __global__ entry points share it — stft_wave_kernel, and
// stft_wave_kernel_occ6 with a register budget for six waves per SIMD, see below)
template <int LOG2_NC, int WAVES, int SHIFT, int OUT, int RES, bool PKV = false, bool SWEEP = false>
__device__ __forceinline__ void stft_wave_body(
    const StftGeom &g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_chan,
    uint32_t n_tiles, const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw, float *__restrict__ minmax,
    uint32_t *__restrict__ queue_head, const WaveOut &wo) {
    using W = WaveFft<LOG2_NC>;
    constexpr int P = W::P, NC = W::NC;
    // SHIFT = -1: "phased" mode for hop = 3 * 128 + 96 samples (the app's 40 ms / 4 at 48 kHz = 480) with n_fft - win
    // >= 96.  |X| does not change when the windowed frame moves inside its zero padding, so every frame is loaded from
    // the 128-sample grid point below its first window sample; its offset from there cycles 0, 96, 64, 32, and the window
    // table (window at offset 0, 48 zero pairs in front) is read that many samples lower.  Consecutive frames then differ
    // by 3, 4, 4, 4 whole register slots: the rotation scheme below with one extra slot move per four frames.  Chunks
    // start on offset-0 frames (host).
    // SHIFT = -2: "dynamic" mode for any other hop in (384, 512) with n_fft - win >= 127 (44.1 kHz: 1764 / 441): the same
    // idea with the offset (any value in [0, 128), odd ones served by a second table whose pairs are shifted by one sample)
    // and the reuse (3 or 4 slots) decided per frame; registers are moved, not rotated (the offsets have no short cycle).
    // (SHIFT = -16 - K encodes the dynamic mode for hops in (128 K, 128 (K + 1)): K or K + 1 slots are reused)
    // (SHIFT = -48 - K: the same when hop and win / 2 are both even — the offset is then always even and the odd table is never read:
    // no second table, which at n_fft 4096 is the eighth wave's LDS)
    constexpr bool PHASED = SHIFT == -1, DYN = SHIFT <= -16, DYN_EVEN = SHIFT <= -48;
    constexpr int DYN_K = DYN_EVEN ? -SHIFT - 48 : DYN ? -SHIFT - 16 : 0;
    static_assert(PHASED || DYN || (SHIFT >= 0 && SHIFT < P), "shift must leave something to reuse");
    static_assert(!PHASED || (P == 16 && OUT != 1), "phased mode: n_fft = 2048, dB output");
    static_assert(!DYN || (DYN_K >= 0 && DYN_K + 1 < P && (OUT != 1 || LOG2_NC == 11)), "dynamic mode: something to reuse; amplitude output at n_fft 4096 only");
    // zero pairs in front of the window table(s): room to read them up to 96 (127) samples lower; DYN: even + odd table
    constexpr int WPAD = PHASED ? 48 : DYN_EVEN ? 0 : DYN ? 64 + NC + 64 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf32 *wtab = reinterpret_cast<cf32 *>(smem_raw);
    // the split-twiddle table is not kept in LDS when the lanes hold their entries in registers (RES bit 3)
    constexpr bool STW_IN_LDS = !((RES & 8) && W::PAIRED);
    cf32 *stw = wtab + NC + WPAD;
    // (the pass-2 constants leave LDS too when the lanes hold them in registers, RES bit 1: at n_fft 4096 those 1280 bytes
    // are what an eighth wave per CU needs)
    constexpr bool T2_IN_LDS = !(RES & 2);
    // (packed pipeline: the split twiddles sit in LDS in pair order, WaveFft::fill_stwp)
    constexpr bool PKP = W::PK && PKV;
    constexpr int STW_LEN = PKP ? W::STWP_LEN : NC;
    cf32 *t2 = stw + (STW_IN_LDS ? STW_LEN : 0);
    cf32 *t3 = t2 + (T2_IN_LDS ? W::T2_LEN : 0);
    cf32 *slabs = t3 + W::T3_LEN;
    uint32_t *meltab = reinterpret_cast<uint32_t *>(slabs + (size_t)WAVES * W::SLAB_LEN);  // OUT == 2 only
    // (the banded mel table's paired layout is read with ds_read_b128, the amplitude row in the slab with ds_read_b64: the
    // table and every slab start on 16 bytes — all lengths in front of them are even numbers of 8-byte entries)
    static_assert((OUT != 2 && OUT != 3) || ((NC + WPAD + (STW_IN_LDS ? STW_LEN : 0) + (T2_IN_LDS ? W::T2_LEN : 0) + W::T3_LEN) % 2 == 0 && W::SLAB_LEN % 2 == 0),
                  "mel table and slabs 16-byte aligned");

    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    // wave index is wave-uniform: tell the compiler, so the frame cursor lives in SGPRs / SALU
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // The wave's first chunk is known from its index alone: look it up and request its first frame before anything else,
    // so that the samples travel while the workgroup fills its LDS tables (a single-track launch is one or two frames per
    // wave: its run time is this start-up).
    // sweep schedule: ticket counter + block ring, 2 * SWEEP_RING + 1 words behind the slabs (the launch adds 128 bytes)
    // (ring slot = one 8-byte word {sequence tag, first chunk}: written and read with single 8-byte DS operations)
    uint64_t *const sw_ring = reinterpret_cast<uint64_t *>(slabs + (size_t)WAVES * W::SLAB_LEN);
    uint32_t &sw_ticket = *reinterpret_cast<uint32_t *>(sw_ring + SWEEP_RING);
    if constexpr (SWEEP) {
        static_assert(!SWEEP || OUT == 0, "sweep: dB rows (the mel table would sit where the scheduler's words are)");
        if (tid == 0) {
            sw_ticket = 0;
            for (uint32_t i = 0; i < SWEEP_RING; i++) sw_ring[i] = 0xffffffffull;
            sw_ring[0] = ((uint64_t)(blockIdx.x * SWEEP_BLK) << 32) | 0u;  // block sequence 0: the workgroup's own index
            sw_ring[1] = ((uint64_t)((atomicAdd(queue_head, 1u) + gridDim.x) * SWEEP_BLK) << 32) | 1u;  // sequence 1
        }
        __syncthreads();
    }
    // next chunk index of this wave (wave-uniform; >= n_tiles: no more work).  All LDS accesses in the explicit address space:
    // through a generic volatile pointer they would be FLAT operations, which count on vmcnt — every draw would wait for the
    // wave's outstanding row stores.  LDS executes one wave's DS operations in order and is the only copy of these words, so
    // program order is all the release / acquire there is (a workgroup-scope fence would drain vmcnt as well).
    typedef volatile __attribute__((address_space(3))) uint64_t lds_u64;
    // The draw in three steps, each issued a frame body ahead of the next so that no LDS / scalar-load latency is exposed:
    //   ticket (LDS atomic, value still in flight)  ->  block slot (opener's duty + ring read)  ->  chunk index
    auto sweep_ticket = [&]() -> uint32_t {  // per-lane value (lane 0 holds the ticket)
        uint32_t t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)&sw_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t;
    };
    auto sweep_slot = [&](uint32_t tk, uint32_t &q, uint32_t &idx) -> uint64_t {
        const uint32_t t = __builtin_amdgcn_readfirstlane(tk);
        q = t / SWEEP_BLK;
        idx = t - q * SWEEP_BLK;
        if (idx == 0 && q >= 1) {  // this wave opens block q: pull the block of sequence q + 1 (even when its own block is empty)
            uint32_t nb = 0;
            if (lane == 0) nb = atomicAdd(queue_head, 1u);
            nb = (__builtin_amdgcn_readfirstlane(nb) + gridDim.x) * SWEEP_BLK;
            if (lane == 0) *(lds_u64 *)&sw_ring[(q + 1) % SWEEP_RING] = ((uint64_t)nb << 32) | (q + 1);
        }
        return *(lds_u64 *)&sw_ring[q % SWEEP_RING];
    };
    auto sweep_chunk = [&](uint64_t slot, uint32_t q, uint32_t idx) -> uint32_t {  // wave-uniform chunk index (>= n_tiles: no more work)
        // (the opener of block q - 1 filled this slot a block's worth of work ago; bounded anyway: a wave that gives up ends)
        for (uint32_t spin = 0; (uint32_t)slot != q; spin++) {
            if (spin > (1u << 24)) {
                // give up LOUDLY: this chunk stays undone, so the launch must not read as a success — word 1 of the
                // queue block is the plan's "sweep dropped a chunk" flag, which the host turns into an error (api.hip: sweep_check)
                if (lane == 0) atomicOr(queue_head + 1, 1u);
                return n_tiles;
            }
            __builtin_amdgcn_s_sleep(2);
            slot = *(lds_u64 *)&sw_ring[q % SWEEP_RING];
        }
        const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(slot >> 32));
        const uint32_t c = base + idx;
        return c < base ? n_tiles : c;  // (overflow guard)
    };
    auto sweep_draw = [&]() -> uint32_t {
        uint32_t q, idx;
        const uint64_t slot = sweep_slot(sweep_ticket(), q, idx);
        return sweep_chunk(slot, q, idx);
    };
    WaveSched sch;
    if constexpr (SWEEP) {
        sch.ahead = 0;
        sch.n_waves = gridDim.x * WAVES;
        sch.use_queue = false;
        sch.armed = false;
        sch.pulled = 0;
        sch.pull_f = 0;
        sch.cur = cursor_at(g, jobs, chunk_tab, n_tiles, sweep_draw());
    } else {
        TH_SCHED_INIT(sch, blockIdx.x * WAVES + wave, TH_PULL_AHEAD);
    }
    cf32 x[P];  // raw samples of the current frame; the frame body refills it for the next one
#define TH_FETCH_FIRST()                                                                                                   \
    do {                                                                                                                   \
        if (sch.cur.valid) {                                                                                               \
            if (sch.cur.edge) /* wave-uniform */                                                                           \
                wave_fetch_reflect<P>(W::lane_col(lane), x, sch.cur.wav, frame_e0(sch.cur, g), sch.cur.n_samples);         \
            else                                                                                                           \
                wave_fetch<P, 0>(W::lane_col(lane), x, sch.cur.wav, frame_e0(sch.cur, g));                                 \
        }                                                                                                                  \
    } while (0)
    TH_FETCH_FIRST();
    // n_fft 4096: the window pairs are stored in the order the lanes read them (lane l reads the pair of column
    // lane_col(l) = (l >> 3) + 8 (l & 7): consecutive lanes would be 64 bytes apart, a 2-way bank conflict on all 32 reads)
    // (the grid-aligned modes read their shifted tables by column: at n_fft 4096 that is the 2-way conflict again, on 32 of
    // the frame's ~250 LDS instructions, against 22 of 30 global loads saved)
    constexpr bool WPERM = (W::PLANES32 || W::PLANES8) && !PHASED && !DYN;  // (n_fft 1024: the same column order)
    static_assert(!WPERM || (!PHASED && !DYN), "the shifted window tables are read by column");
    for (uint32_t i = tid; i < NC; i += 64 * WAVES) {
        const uint32_t col = i & 63u, li = WPERM ? (i & ~63u) + 8u * (col & 7u) + (col >> 3) : i;
        wtab[WPAD + li] = wtab_g[(DYN_EVEN ? 64 : WPAD) + i];  // (even-only dynamic mode: the even table behind the global buffer's 64 zero pairs)
        if constexpr (STW_IN_LDS && !PKP) stw[i] = tw[i];
    }
    if constexpr (STW_IN_LDS && PKP) W::fill_stwp(tid, 64 * WAVES, tw, stw);
    for (uint32_t i = tid; i < WPAD; i += 64 * WAVES) wtab[i] = wtab_g[i];
    W::fill_tables(tid, 64 * WAVES, tw, T2_IN_LDS ? t2 : nullptr, t3);
    if constexpr ((OUT == 2 || OUT == 3) && LOG2_NC != 11)  // (n_fft 4096 keeps no table in LDS: the moment form, mel_moments_global)
        for (uint32_t i = tid; i < wo.mel_words; i += 64 * WAVES) meltab[i] = wo.mel_tab[i];
    __syncthreads();

    cf32 *slab = slabs + (size_t)wave * W::SLAB_LEN;
    // fused mel at n_fft 1024: a (r, f) buffer of MEL_PRF_1024 pieces per wave behind the mel table (see wave_frame)
    cf32 *mel_prf = nullptr;
    if constexpr (OUT == 2 && LOG2_NC == 9)
        if (wo.mel_slots != 0)  // (the banded sums have no partial-sum buffer: the launch does not allocate it)
            mel_prf = reinterpret_cast<cf32 *>(meltab + ((wo.mel_words + 1u) & ~1u)) + (size_t)wave * MEL_PRF_1024;
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    const uint32_t lane_wave = lane;
    // Register rotation instead of register moves: with hop = SHIFT slots, frame f+1 is frame f moved down by
    // SHIFT slots.  Rather than moving P - SHIFT complex registers per frame, the frame body is instantiated
    // NROT = P/SHIFT times; instance ROT reads logical slot m from physical x[(m + ROT*SHIFT) % P] and loads
    // the next frame's SHIFT new slots over the ones that just went out of the window.  A new chunk (all P
    // slots loaded) always starts at rotation 0.
    constexpr int SHIFT_NZ = SHIFT > 0 ? SHIFT : 1;  // (keeps the constant expression below free of a % 0)
    constexpr bool ROTATE = PHASED || (SHIFT > 0 && P % SHIFT_NZ == 0 && P / SHIFT_NZ <= 4);
    constexpr int NROT = PHASED ? 4 : (ROTATE ? P / SHIFT : 1);
    constexpr int RESK = (PHASED || DYN || SWEEP) ? (RES & ~1) : RES;  // phased: the window changes every frame, it stays in LDS; sweep: 148 VGPRs instead of 168 + scratch
    // slots reused / rotation offset / window table of body ROT
#define TH_BODY_SHIFT(ROT) (PHASED ? ((ROT) == 0 ? 3 : 4) : DYN ? DYN_K : SHIFT)
#define TH_BODY_OFF(ROT) (PHASED ? ((ROT) == 0 ? 0 : 4 * (ROT) - 1) : (ROTATE ? (ROT) * SHIFT : 0))
#define TH_FRAME_P(ROT, MELP)                                                                                          \
    wave_frame<LOG2_NC, TH_BODY_SHIFT(ROT), OUT, ROTATE, TH_BODY_OFF(ROT), RESK, PHASED ? (ROT) : DYN_EVEN ? -3 : DYN ? -2 : -1, PKV, false, (MELP)>( \
        g, wtab + (PHASED ? WPAD - ((96 * (ROT)) & 127) / 2 : 0), stw, t2, t3, slab, lane_wave, f, cur.f1, cur.wav, cur.n_samples, cur.spec, cur.spec_pitch, x, rw, rw2, rwa, rwb, rws, rw_mid, lmin, \
        lmax, meltab, mel_prf, wo, amp_a)
#define TH_FRAME(ROT) TH_FRAME_P(ROT, 0)
    float amp_a[OUT == 3 ? W::N_EMIT : 1];  // frame pairs: the first frame's amplitudes (wave_frame, MELP)
    // per-lane constant tables kept in registers for the whole launch (see wave_frame)
    cf32 rw[(RESK & 1) ? P : 1], rw2[(RESK & 2) ? W::NT2 : 1];
    cf32 rwa[(RESK & 4) ? W::NQ : 1][W::NT3], rwb[(RESK & 4) ? W::NQ : 1][W::NT3], rws[(RESK & 8) ? W::NQ : 1][W::R3];
    if constexpr (RESK & 1) {
#pragma unroll
        for (int m = 0; m < P; m++) rw[m] = wtab[(WPERM ? lane : W::lane_col(lane)) + 64u * m];
    }
    if constexpr (RESK & 2) W::load_t2_from_tw(lane, rw2, tw);
    if constexpr (RESK & 4) W::load_t3_paired(lane, rwa, rwb, t3);
    cf32 rw_mid = {0.0f, 0.0f};
    if constexpr ((RESK & 8) != 0 && W::PAIRED) {  // straight from the global table: no LDS copy exists
#pragma unroll
        for (int q = 0; q < W::NQ; q++)
#pragma unroll
            for (int s = 0; s < W::R3; s++) rws[q][s] = tw[W::split_k(lane, q, s)];
        rw_mid = tw[NC / 2];
    }
    if constexpr (PKP) rw_mid = tw[NC / 2];  // (the pair-ordered LDS table has no entry for the self-mirrored bin)
    if constexpr (SWEEP) {
        static_assert(!SWEEP || (SHIFT > 0 && P / SHIFT_NZ == 4 && P % SHIFT_NZ == 0), "sweep: rotation mode with four bodies");
#define TH_FRAME_SW(ROT, FULL)                                                                                         \
    wave_frame<LOG2_NC, SHIFT, OUT, true, (ROT) * SHIFT, RESK, -1, PKV, FULL>(                                          \
        g, wtab, stw, t2, t3, slab, lane_wave, f, cur.f1, cur.wav, cur.n_samples, cur.spec, cur.spec_pitch, x, rw, rw2, rwa, rwb, rws, rw_mid, lmin, \
        lmax, meltab, mel_prf, wo, amp_a, true, nwav, ne0)
#define TH_SW_NEXT() (nxt = cursor_next(g, jobs, chunk_tab, n_tiles, sweep_draw(), cur))
        while (sch.cur.valid) {
            const FrameCursor cur = sch.cur;
            lmin = __builtin_inff();
            lmax = -__builtin_inff();
            uint32_t f = cur.f;
            const uint32_t n = cur.f1 - cur.f;  // 4 frames; fewer at the tail of a channel, 1 for a boundary frame
            FrameCursor nxt{};
            bool pre = false;  // the next chunk's first frame has been requested by this chunk's last frame
            gptr<const float> nwav = cur.wav;
            int64_t ne0 = 0;
            if (n == 4) {
                // the next chunk is drawn in steps between the frame bodies (ticket | block slot | chunk-table entry | cursor):
                // every step's LDS / scalar-load latency hides behind a whole frame
                const uint32_t tk = sweep_ticket();
                TH_FRAME_SW(0, false);
                ++f;
                uint32_t sq, sidx;
                const uint64_t slot = sweep_slot(tk, sq, sidx);
                TH_FRAME_SW(1, false);
                ++f;
                nxt = cursor_next(g, jobs, chunk_tab, n_tiles, sweep_chunk(slot, sq, sidx), cur);
                pre = nxt.valid && !nxt.edge;
                if (pre) {
                    nwav = nxt.wav;
                    ne0 = frame_e0(nxt, g);
                } else {  // nothing to prefetch: the last frame re-reads its own span (in bounds; the result is never used)
                    ne0 = frame_e0(cur, g) + 3 * (int64_t)g.hop;
                }
                TH_FRAME_SW(2, false);
                ++f;
                TH_FRAME_SW(3, true);
            }
            else {  // tail of a channel (1 .. 3 frames) or a boundary frame: ONE more body (every frame loaded in full) instead of
                // three rotation bodies — 53 -> 41 KB of code, and with the window read from LDS (RESK) no register spills: the
                // first version kept two per-lane words in scratch and reloaded them in every frame body, each reload behind
                // the previous frame's row stores (one in-order vmcnt)
                for (;;) {
                    wave_frame<LOG2_NC, 0, OUT, false, 0, RESK, -1, PKV>(
                        g, wtab, stw, t2, t3, slab, lane_wave, f, cur.f1, cur.wav, cur.n_samples, cur.spec, cur.spec_pitch, x, rw, rw2, rwa, rwb, rws, rw_mid, lmin,
                        lmax, meltab, mel_prf, wo, amp_a);
                    if (++f >= cur.f1) break;
                }
                TH_SW_NEXT();
            }
            if (minmax != nullptr) {
                const float a = wave_min(lmin), b = wave_max(lmax);
                if (lane == 0) {
                    minmax[2 * (size_t)cur.t] = a;
                    minmax[2 * (size_t)cur.t + 1] = b;
                }
            }
            sch.cur = nxt;
            if (!pre) TH_FETCH_FIRST();
        }
#undef TH_FRAME_SW
#undef TH_SW_NEXT
        return;
    }
    // chunk loop: one queue pull and one full fetch per chunk of up to frames_per_tile consecutive frames
    while (sch.cur.valid) {
        const FrameCursor &cur = sch.cur;
        lmin = __builtin_inff();
        lmax = -__builtin_inff();
        uint32_t f = cur.f;
        // frame loop (steady state: branch-free register flow, see wave_frame)
        // (The instruction arbiter serves the oldest wave of a SIMD first, and strictly: of the three waves of a SIMD the
        // first-launched runs a frame in 4.4 us, the second in 5.5, the third in 8.1 (scripts/wave_times.py).  Cycling
        // s_setprio 0..NROT-1 from frame to frame made all waves advance at the same pace (5.4 / 5.7 / 6.0 us) — and the
        // launch 1.5 % slower: the sum of the rates is what counts, the dynamic chunk queue already absorbs the different
        // speeds, and the kernel is bound by its total work, not by its tail.  Measured in round 2, not kept.)
        if constexpr (OUT == 3) {
            // frame pairs: bodies alternate first / second frame (rotation index as below); a chunk that ends on a first frame
            // is finished by wave_mel_flush
            static_assert(OUT != 3 || (NROT == 1 || NROT == 2 || NROT == 4), "frame pairs: an even number of frame bodies per cycle");
            bool pending;
            for (;;) {
                TH_SCHED_PULL(sch, f, lane);
                TH_FRAME_P(0, OUT == 3 ? 1 : 0);
                pending = true;
                if (++f >= cur.f1) break;
                TH_SCHED_PULL(sch, f, lane);
                TH_FRAME_P(NROT > 1 ? 1 : 0, OUT == 3 ? 2 : 0);
                pending = false;
                if (++f >= cur.f1) break;
                if constexpr (NROT > 2) {
                    TH_SCHED_PULL(sch, f, lane);
                    TH_FRAME_P(NROT > 2 ? 2 : 0, OUT == 3 ? 1 : 0);
                    pending = true;
                    if (++f >= cur.f1) break;
                    TH_SCHED_PULL(sch, f, lane);
                    TH_FRAME_P(NROT > 2 ? 3 : 0, OUT == 3 ? 2 : 0);
                    pending = false;
                    if (++f >= cur.f1) break;
                }
                if constexpr (PHASED) {  // (as below: one slot move restores rotation 0)
                    const cf32 t = x[P - 1];
#pragma unroll
                    for (int m = P - 1; m > 0; m--) x[m] = x[m - 1];
                    x[0] = t;
                }
            }
            if constexpr (OUT == 3)
                if (pending) wave_mel_flush<LOG2_NC>(slab, lane_wave, f - 1, cur.spec, cur.spec_pitch, amp_a, lmin, lmax, meltab, wo);
        } else
        for (;;) {
            TH_SCHED_PULL(sch, f, lane);
            TH_FRAME(0);
            if (++f >= cur.f1) break;
            if constexpr (NROT > 1) {
                TH_SCHED_PULL(sch, f, lane);
                TH_FRAME(1);
                if (++f >= cur.f1) break;
            }
            if constexpr (NROT > 2) {
                TH_SCHED_PULL(sch, f, lane);
                TH_FRAME(2);
                if (++f >= cur.f1) break;
            }
            if constexpr (NROT > 3) {
                TH_SCHED_PULL(sch, f, lane);
                TH_FRAME(3);
                if (++f >= cur.f1) break;
            }
            if constexpr (PHASED) {  // 3 + 4 + 4 + 4 = 15 slots in four frames: one slot move restores rotation 0
                const cf32 t = x[P - 1];
#pragma unroll
                for (int m = P - 1; m > 0; m--) x[m] = x[m - 1];
                x[0] = t;
            }
        }
        // (min, max) of this chunk: a plain store per chunk, folded per channel by wave_post_kernel afterwards.
        // (Float atomics on the channel's slot from every wave are served one at a time, ~8 ns each: 50 us for the
        // 2800 waves of a single-track launch whose FFT work takes 20.)
        if (minmax != nullptr) {
            const float a = wave_min(lmin), b = wave_max(lmax);
            if (lane == 0) {
                minmax[2 * (size_t)cur.t] = a;
                minmax[2 * (size_t)cur.t + 1] = b;
            }
        }
        TH_SCHED_ADVANCE(sch, lane);
        TH_FETCH_FIRST();  // the next chunk's first frame
    }
#undef TH_FETCH_FIRST
#undef TH_FRAME
#undef TH_FRAME_P
#undef TH_BODY_SHIFT
#undef TH_BODY_OFF
}

template <int LOG2_NC, int WAVES, int SHIFT, int OUT, int RES, bool PKV = false, bool SWEEP = false>
__global__ __launch_bounds__(64 * WAVES) void stft_wave_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_chan,
    uint32_t n_tiles, const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw, float *__restrict__ minmax,
    uint32_t *__restrict__ queue_head, WaveOut wo) {
    stft_wave_body<LOG2_NC, WAVES, SHIFT, OUT, RES, PKV, SWEEP>(g, jobs, chunk_tab, n_chan, n_tiles, wtab_g, tw, minmax, queue_head, wo);
}
// The same with at least six waves per SIMD asked of the register allocator (HIP's second __launch_bounds__ argument: 80 VGPRs).
// The n_fft 1024 kernel runs two workgroups of twelve waves per CU; its frame-pair mel variant keeps nine amplitudes across the
// next frame's transform and would otherwise take 81 registers — one more than two workgroups leave a wave.
template <int LOG2_NC, int WAVES, int SHIFT, int OUT, int RES>
__global__ __launch_bounds__(64 * WAVES, 6) void stft_wave_kernel_occ6(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_chan,
    uint32_t n_tiles, const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw, float *__restrict__ minmax,
    uint32_t *__restrict__ queue_head, WaveOut wo) {
    stft_wave_body<LOG2_NC, WAVES, SHIFT, OUT, RES, false, false>(g, jobs, chunk_tab, n_chan, n_tiles, wtab_g, tw, minmax, queue_head, wo);
}


#endif  // !TH_PART_MAIN
#if TH_PART_MAIN  // multi-frame and workgroup-per-frame kernels, launchers, host-side helpers
// ------------------------------------------------------------------------------------------
// Multi-frame wave kernel (stft_wave_multi.h): n_fft = 1024 -> two frames per wave, n_fft = 512 -> four; every lane
// holds 16 complex points, the instruction stream is that of the 2048-point plan.  Same launch shape, chunk queue, job
// tables and (min, max) protocol as stft_wave_kernel.  Group g of the lanes takes frame f + g of the wave's chunk; in
// the last iteration of a chunk the groups past its end recompute (and re-store, identically) the chunk's last frame,
// so control flow, LDS traffic and stores stay wave-uniform.  Every frame is loaded in full (no register reuse across
// iterations: a group advances by G hops); the loads for the next iteration are issued right after the window multiply
// and land during the transform.  dB output only (mel plans at these sizes use the one-frame kernel / generic kernel).
// NLD > 0 (n_fft 512, even hop <= 256): the G frames of an iteration overlap, and every group loading its own copy made
// the kernel issue 4.6x the load instructions of the 2048 plan for the same audio (PMC, round 2).  The wave instead loads
// the (G - 1) hop + n_fft samples the iteration spans ONCE — NLD 16-byte loads per lane, 1 KB per instruction, into
// NLD * 4 registers instead of 32 — and at the start of the next iteration passes them through its (free) exchange slab:
// ds_write_b128, then each lane reads the 16 sample pairs of its own frame.
// ------------------------------------------------------------------------------------------
// OUT: 0 dB rows, 1 amplitude rows (first half of the two-kernel mel paths), 2 mel rows (n_fft 512 under narrow filters: the
// banded sums of mel_rows_kernel as an epilogue — the four frames' amplitudes go to the wave's slab instead of memory, then
// lane = mel: per group of 64 mels the lane's first bin and 8 weights come from the table in LDS once, and each of the four
// frames costs 8 LDS reads + 8 FMAs + one 256-byte row store; table layout as launch_mel_rows, kernels.h)
constexpr int MELR_AP = 272;  // floats per amplitude row in the slab (257 bins + the 8-bin reach of the last filters, 16-bank skew between the groups)
template <int LOG2_NC, int WAVES, int OUT, int NLD>
__global__ __launch_bounds__(64 * WAVES) void stft_wave_multi_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_chan,
    uint32_t n_tiles, const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw, float *__restrict__ minmax,
    uint32_t *__restrict__ queue_head, WaveOut wo) {
    using W = WaveFftM<LOG2_NC>;
    constexpr int P = W::P, NC = W::NC, G = W::G, L = W::L;
    constexpr bool AMP = OUT == 1, MELR = OUT == 2 || OUT == 4, MELM = OUT == 4;  // (4: mel rows in the moment form, table in global memory — filters wider than the banded table's 8 bins)
    static_assert(!MELR || (LOG2_NC == 8 && 4 * MELR_AP * (int)sizeof(float) <= (int)(sizeof(cf32) * W::SLAB_LEN) &&
                            NC + MEL_ROWS_W <= MELR_AP), "mel rows: four amplitude rows in the wave's slab");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf32 *wtab = reinterpret_cast<cf32 *>(smem_raw);
    cf32 *stw = wtab + NC;
    cf32 *t2 = stw + NC;
    cf32 *t3 = t2 + W::T2_LEN;
    cf32 *slabs = t3 + W::T3_LEN;
    uint32_t *meltab = reinterpret_cast<uint32_t *>(slabs + (size_t)WAVES * W::SLAB_LEN);  // OUT == 2 only
    const uint32_t tid = threadIdx.x;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < NC; i += 64 * WAVES) {
        wtab[i] = wtab_g[i];
        stw[i] = tw[i];
    }
    W::fill_tables(tid, 64 * WAVES, tw, t2, t3);
    if constexpr (MELR)
        for (uint32_t i = tid; i < wo.mel_words; i += 64 * WAVES) meltab[i] = wo.mel_tab[i];
    __syncthreads();
    cf32 *slab = slabs + (size_t)wave * W::SLAB_LEN;
    const uint32_t lane_wave = tid & 63u;
    const cf32 w_mid = stw[NC / 2];
    WaveSched sch;
    TH_SCHED_INIT(sch, blockIdx.x * WAVES + wave, (uint32_t)G > TH_PULL_AHEAD ? (uint32_t)G : TH_PULL_AHEAD);
    while (sch.cur.valid) {
        const FrameCursor &cur = sch.cur;
        float lmin = __builtin_inff(), lmax = -__builtin_inff();
        const int32_t lead = (int32_t)(g.win / 2 + g.pad_left);
        // samples of the lane's frame: x[m] = (s[2 n], s[2 n + 1]), n = col + L m, s = the frame's n_fft-sample span.
        // Wave-uniform base (SGPRs) + a small per-lane offset: the group's frame starts dg hops further.  The span is clamped
        // into the channel: a no-op for interior frames; the one-frame chunks of boundary frames "prefetch themselves" in
        // the loop below, that span is partly outside, and the result is never used.
        auto fetch = [&](cf32(&x)[P], uint32_t f_it, uint32_t lane) {
            const uint32_t grp = W::grp(lane), col = W::lane_col(lane);
            const uint32_t last = cur.f1 - 1u - f_it;  // groups past the chunk's end repeat its last frame
            const uint32_t dg = grp < last ? grp : last;
            const int32_t e0_max = (int32_t)cur.n_samples - (int32_t)g.n_fft;
            int32_t e0 = (int32_t)((int64_t)(f_it + dg) * g.hop) - lead;
            e0 = e0 < 0 ? 0 : (e0 > e0_max ? e0_max : e0);
            const gptr<const float> base = cur.wav + (uint32_t)e0 + 2u * col;
#pragma unroll
            for (int m = 0; m < P; m++) {
                const gptr<const float> p = base + 2u * (uint32_t)L * m;
                x[m] = {p[0], p[1]};
            }
        };
        // NLD > 0: the samples from the first frame's start on, lane i holds samples 4 i + 256 m .. + 3 of that span in raw[m]
        // (indices clamped into the channel: beyond the span the iteration needs, and never used there)
        typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));  // global side: any 4-byte aligned span
        typedef float f32x4a __attribute__((ext_vector_type(4)));              // LDS side: 16-byte aligned
        constexpr int NRAW = NLD > 0 ? NLD : 1;
        f32x4u raw[NRAW];
        auto fetch_raw = [&](uint32_t f_it, uint32_t lane) {
            int32_t e0 = (int32_t)((int64_t)f_it * g.hop) - lead;
            e0 = e0 < 0 ? 0 : e0;
            const uint32_t lim = cur.n_samples - 4u;
#pragma unroll
            for (int m = 0; m < NRAW; m++) {
                uint32_t idx = (uint32_t)e0 + 4u * lane + 256u * (uint32_t)m;
                idx = idx < lim ? idx : lim;
                raw[m] = *(gptr<const f32x4u>)(cur.wav + idx);
            }
        };
        cf32 x[P];
        if (cur.edge) {  // wave-uniform: boundary frame (one-frame chunk), numpy-'reflect' indexing per sample
            const uint32_t col = W::lane_col(lane_wave);
            const int32_t e0 = (int32_t)((int64_t)cur.f * g.hop) - lead;
#pragma unroll
            for (int m = 0; m < P; m++) {
                const int32_t i = e0 + 2 * (int32_t)(col + (uint32_t)L * m);
                x[m] = {cur.wav[reflect_once(i, (int32_t)cur.n_samples)], cur.wav[reflect_once(i + 1, (int32_t)cur.n_samples)]};
            }
        } else if constexpr (NLD > 0) {
            fetch_raw(cur.f, lane_wave);
        } else {
            fetch(x, cur.f, lane_wave);
        }
        for (uint32_t f = cur.f; f < cur.f1; f += G) {
            TH_SCHED_PULL(sch, f, lane_wave);
            uint32_t lane = lane_wave;
            asm volatile("" : "+v"(lane));  // per-iteration copy: keeps the lane-derived addresses out of the loop-invariant set
            lane &= 63u;
            cf32 z[P];
            if constexpr (NLD > 0) {
                if (!cur.edge) {  // wave-uniform: the iteration's samples through the slab (free since the last read of exchange 2)
                    float *const stage = reinterpret_cast<float *>(slab);
#pragma unroll
                    for (int m = 0; m < NRAW; m++)
                        *(__attribute__((address_space(3))) f32x4a *)(stage + 4u * lane + 256u * (uint32_t)m) = (f32x4a)raw[m];
                    wave_lds_sync();
                    const uint32_t grp = W::grp(lane), last = cur.f1 - 1u - f, dg = grp < last ? grp : last;
                    const cf32 *const src = reinterpret_cast<const cf32 *>(stage + dg * g.hop) + W::lane_col(lane);
#pragma unroll
                    for (int m = 0; m < P; m++) x[m] = lds_ld(&src[(uint32_t)L * m]);
                    wave_lds_sync();  // pass 1 rewrites the slab
                }
            }
            {   // window pairs from LDS (slot m = complex point col + L m of the group's frame): one read serves G frames'
                // worth of lanes, and keeping them in registers instead would push the kernel over its 168 VGPRs
                const uint32_t col = W::lane_col(lane);
                cf32 w[P];
#pragma unroll
                for (int m = 0; m < P; m++) w[m] = lds_ld(&wtab[col + (uint32_t)L * m]);
#pragma unroll
                for (int m = 0; m < P; m++) z[m] = {x[m].re * w[m].re, x[m].im * w[m].im};
            }
            {   // the next iteration's frames now (branch-free: the last iteration re-reads its own)
                const uint32_t fn = f + G < cur.f1 ? f + G : f;
                if constexpr (NLD > 0) fetch_raw(fn, lane);
                else fetch(x, fn, lane);
            }
            TH_SCHED_BARRIER();
            cf32 w2[W::NT2];
            W::load_t2(lane, w2, t2);
            W::pass1(lane, z, slab);
            wave_lds_sync();
            TH_SCHED_BARRIER();
            W::read1(lane, z, slab);
            wave_lds_sync();
            W::pass2_w(lane, z, w2, slab);
            wave_lds_sync();
            TH_SCHED_BARRIER();
            cf32 wa[W::NW3], wb[W::NW3];
            W::load_t3_paired(lane, wa, wb, t3);
            cf32 za[W::NQ][W::R3], zb[W::NQ][W::R3];
            W::read2_paired(lane, za, zb, slab);
            wave_lds_sync();  // the slab is free again
            cf32 ws[W::NQ][W::R3];
            W::load_stw_paired(lane, ws, stw);
            TH_SCHED_BARRIER();
            W::pass3_paired_w(za, zb, wa, wb);
            TH_SCHED_BARRIER();
            const uint32_t grp = W::grp(lane), last = cur.f1 - 1u - f, dg = grp < last ? grp : last;
            const gptr<float> row = cur.spec + (size_t)f * cur.spec_pitch + (size_t)dg * cur.spec_pitch;
            float *const amp_row = reinterpret_cast<float *>(slab) + grp * (uint32_t)MELR_AP;  // OUT == 2: the group's amplitude row
            W::split_paired_w(lane, za, zb, ws, w_mid, [&](uint32_t kb, int kc, float p) {
                const gptr<float> dst = (gptr<float>)((gptr<char>)row + (uint64_t)(kb << 2) + (int64_t)(4 * kc));  // base + immediate (split_base)
                if constexpr (MELR) {  // (pre-scaled by 2^32 like the fused epilogue of the one-frame kernels: amp_to_dB_fast)
                    amp_row[kb + (uint32_t)kc] = power_to_amp_scaled(p);
                } else if constexpr (AMP) {  // amplitude rows for the matrix-core mel path (spectrogram.rs:200-207)
                    *dst = power_to_amp(p);
                } else {
                    const float d = power_to_dB(p);
                    *dst = d;
                    lmin = nmin(lmin, d);
                    lmax = nmax(lmax, d);
                }
            });
            if constexpr (MELR) {
                // the filters' reach past the last bin: zero (weights there are zero, whatever the slab held must not be NaN)
                if (W::lig(lane) < (uint32_t)(MELR_AP - NC - 1)) amp_row[NC + 1 + W::lig(lane)] = 0.0f;
                wave_lds_sync();
                const float *const ampf = reinterpret_cast<const float *>(slab);
                constexpr int MW = MEL_ROWS_W;
                const uint32_t height = wo.n_mel, pad = cur.spec_pitch - height;
                const uint32_t npad = (pad < 32u && cur.spec_pitch % 32u == 0) ? pad : 0u;
                if constexpr (MELM) {  // filters wider than the banded table's 8 bins — the moment form, frame after frame (table in global memory)
#pragma unroll 1
                    for (int fr = 0; fr < G; fr++) {
                        const uint32_t dgf = (uint32_t)fr < last ? (uint32_t)fr : last;  // (groups past the chunk's end repeat its last frame)
                        const gptr<float> orow = cur.spec + (size_t)f * cur.spec_pitch + (size_t)dgf * cur.spec_pitch;
                        mel_moments_global(lane, ampf + fr * MELR_AP, as_global(wo.mel_tab), wo.mel_groups, [&](uint32_t m, float v) {
                            if (m < height) {
                                const float d = amp_to_dB_fast(v);
                                orow[m] = d;
                                lmin = nmin(lmin, d);
                                lmax = nmax(lmax, d);
                            }
                        });
                        if (lane < npad) orow[height + lane] = 0.0f;  // complete the row's last 128-byte line (see wave_frame)
                    }
                } else
#pragma unroll
                for (int gq = 0; gq < MEL_ROWS_MAX_GROUPS; gq++) {
                    if ((uint32_t)gq < wo.mel_groups) {  // wave-uniform
                        const uint32_t *const tg = meltab + (uint32_t)gq * (MW + 1) * 64u + lane;
                        const uint32_t lo = tg[0];
                        // (explicit LDS pointers: every read below is one running base + an immediate offset, see mel_banded)
                        TH_LDS_F32 *const wp = TH_LDS_F32_PTR(reinterpret_cast<const float *>(tg));
                        TH_LDS_F32 *const ap = TH_LDS_F32_PTR(ampf + lo);
                        float wq[MW];
#pragma unroll
                        for (int t = 0; t < MW; t++) wq[t] = wp[(1 + t) * 64];
                        const uint32_t m = 64u * gq + lane;
#pragma unroll
                        for (int fr = 0; fr < G; fr++) {
                            float a[MW];
#pragma unroll
                            for (int t = 0; t < MW; t++) a[t] = ap[fr * MELR_AP + t];
                            float acc = a[0] * wq[0];
#pragma unroll
                            for (int t = 1; t < MW; t++) acc = __builtin_fmaf(a[t], wq[t], acc);
                            const uint32_t dgf = (uint32_t)fr < last ? (uint32_t)fr : last;  // (groups past the chunk's end repeat its last frame)
                            const gptr<float> orow = cur.spec + (size_t)f * cur.spec_pitch + (size_t)dgf * cur.spec_pitch;
                            if (m < height) {
                                const float d = amp_to_dB_fast(acc);
                                orow[m] = d;
                                lmin = nmin(lmin, d);
                                lmax = nmax(lmax, d);
                            } else if (m - height < npad) {
                                orow[m] = 0.0f;  // complete the row's last 128-byte line (see wave_frame)
                            }
                        }
                    }
                }
                wave_lds_sync();  // the next iteration stages its samples in the slab
            } else {   // complete the row's last 128-byte line (see wave_frame)
                const uint32_t height = (uint32_t)(NC + 1), pad = cur.spec_pitch - height, l = W::lig(lane);
                if (l - 1u < ((pad < 32u && cur.spec_pitch % 32u == 0) ? pad : 0u)) row[height - 1u + l] = 0.0f;
            }
            TH_SCHED_BARRIER();
        }
        if (minmax != nullptr) {
            const float a = wave_min(lmin), b = wave_max(lmax);
            if (lane_wave == 0) {
                minmax[2 * (size_t)cur.t] = a;
                minmax[2 * (size_t)cur.t + 1] = b;
            }
        }
        TH_SCHED_ADVANCE(sch, lane_wave);
    }
}

template <int LOG2_NC, int WAVES, int OUT, int NLD>
static hipError_t launch_wave_multi_n(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan,
                                      uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax,
                                      uint32_t *d_queue_head, uint32_t n_cu, const WaveOut &out, hipStream_t s) {
    using W = WaveFftM<LOG2_NC>;
    auto kern = stft_wave_multi_kernel<LOG2_NC, WAVES, OUT, NLD>;
    static_assert(NLD * 1024 <= (int)(sizeof(cf32) * W::SLAB_LEN), "the staged samples fit the wave's slab");
    if (OUT == 4 ? (out.mel_tab == nullptr || out.mel_groups == 0 || out.mel_words != 0 || out.mel_moment == 0)
                 : (OUT == 2 && (out.mel_tab == nullptr || out.mel_groups == 0 || out.mel_groups > (uint32_t)MEL_ROWS_MAX_GROUPS ||
                                 out.mel_words != out.mel_groups * (MEL_ROWS_W + 1) * 64u)))
        return hipErrorInvalidValue;
    const size_t lds = sizeof(cf32) * ((size_t)2 * W::NC + W::T2_LEN + W::T3_LEN + (size_t)WAVES * W::SLAB_LEN) +
                       (OUT == 2 ? (size_t)out.mel_words * 4 : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint32_t wg_needed = (n_tiles + WAVES - 1) / WAVES;
    const uint32_t per_cu = 2 * lds <= 160 * 1024 ? 2u : 1u;  // (8 waves: two independent workgroups per CU)
    const uint32_t grid = wg_needed < n_cu * per_cu ? wg_needed : n_cu * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, s, g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw,
                       d_minmax, d_queue_head, out);
    return hipGetLastError();
}
// staged loads (see the kernel): n_fft 512, even hop (8-byte LDS reads), span (G - 1) hop + n_fft within NLD KB of samples
template <int LOG2_NC, int WAVES, int OUT>
static hipError_t launch_wave_multi_t(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan,
                                      uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax,
                                      uint32_t *d_queue_head, uint32_t n_cu, const WaveOut &out, hipStream_t s) {
    if constexpr (LOG2_NC == 8) {
        const uint32_t span = 3u * g.hop + g.n_fft;
        if (g.hop % 2u == 0 && span <= 1024u)
            return launch_wave_multi_n<LOG2_NC, WAVES, OUT, 4>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
        if (g.hop % 2u == 0 && span <= 1280u)
            return launch_wave_multi_n<LOG2_NC, WAVES, OUT, 5>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
    }
    return launch_wave_multi_n<LOG2_NC, WAVES, OUT, 0>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
}
template <int LOG2_NC, int OUT>
static hipError_t launch_wave_multi(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan,
                                    uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax,
                                    uint32_t *d_queue_head, uint32_t n_cu, int waves, const WaveOut &out, hipStream_t s) {
    switch (waves <= 0 ? 12 : waves) {
        case 8: return launch_wave_multi_t<LOG2_NC, 8, OUT>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
        case 12: return launch_wave_multi_t<LOG2_NC, 12, OUT>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
        case 16: return launch_wave_multi_t<LOG2_NC, 16, OUT>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------
// Block kernel (stft_block.h): n_fft = 8192 / 16384, one workgroup of T = Nc / 16 threads per frame, 16 points per thread,
// radix-16 passes in registers with LDS exchanges between them.  One workgroup per chunk of frames_per_tile consecutive
// interior frames (chunk table as for the wave kernels; the hardware dispatcher balances the workgroups); boundary frames
// and channels shorter than n_fft go to the generic kernel.  Per-pass twiddle constants (10 per pass, FMA butterflies)
// and, for n_fft 8192, the thread's 16 window pairs live in registers for the whole chunk.  Output: dB rows (linear scale)
// or amplitude rows (AMP: first half of the matrix-core mel path).
// Full-reload variant 167 / 196 VGPRs (three workgroups per CU at n_fft 8192, one at 16384), reuse variant 198 / 216 (two /
// one): 8192/2048 0.78 -> 0.72 ms, 16384/4096 1.32 -> 1.11 ms.  Measured and not adopted:
// the mid-pass constants from LDS tables, the window from global memory at 8192 (both within +-3 %), forcing four waves
// per SIMD with amdgpu_waves_per_eu (128 VGPRs, 140-240 bytes of scratch: 0.77 -> 1.32 ms and 1.33 -> 1.62 ms), requesting
// the next frame's samples during the split pass (32 more VGPRs: n_fft 16384 +-0, 8192 loses its third workgroup: 0.78 -> 0.93 ms).
// ------------------------------------------------------------------------------------------
// REUSE (hop = n_fft / 4): frame f + 1 is frame f moved by four of the thread's 16 slots, so the raw samples stay in
// registers, move down four slots and only the new hop is loaded — requested right behind the window multiply, a whole
// transform ahead of its use.  Without it every frame is loaded in full and three quarters of that are re-reads that the
// L2 (4 MB per XCD against 96 workgroups streaming 48 KB per frame each) mostly misses: PMC 1.87 GB fetched per launch for
// 0.74 GB of audio.  The 32 registers come from the window pairs, which are read from the (L2-resident) table instead.
// VT (round 4): "virtual threads" per thread.  With VT = 2 a workgroup has T / 2 threads and every thread runs the lane
// functions of BlockFft for the two thread ids t and t + T / 2 (stft_block.h is unchanged: its functions take t as an argument):
// twice the independent work between two barriers, half the waves at each barrier, 256 registers per thread; a pass's
// constants are requested behind the previous pass's LDS stores (the constants of a pass with sub-size NS <= T / 2 depend on
// t mod NS only: both virtual threads share them).  Measured (profiles/r04_ab_block_virtual_threads.txt):
//   n_fft 32768: 1.94 ms against 2.19 (hop = n_fft / 4), 3.26 against 3.69 (19200 / 4800): the default there.  With the raw
//                samples resident on top (REUSE: 256 VGPRs + 120 bytes of scratch) 1.98: not used.
//   n_fft 16384: ONE exchange buffer per workgroup of four waves, so that two workgroups share a CU: 1.35 ms against 1.01 for
//                one workgroup of eight waves with two buffers and resident constants: not used.
// Row stores of the block kernels: non-temporal (round 5, as in stft_subwave_kernel: the rows a frame writes should not push the
// samples and window pairs the next frame re-reads out of the L2).  -DTH_BLOCK_NT=0: A/B builds.
#if !defined(TH_BLOCK_NT)
#define TH_BLOCK_NT 1
#endif
#if TH_BLOCK_NT
#define TH_BLOCK_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define TH_BLOCK_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
constexpr bool block_double_buffered(int log2_nc, int vt) { return log2_nc == 13 && vt == 1; }
// OUT: 0 dB rows, 1 amplitude rows (first half of the two-kernel mel paths), 2 mel rows (round 6): the amplitudes go to the exchange
// buffer that is free behind the split pass, and the filterbank is applied in its moment form (mel_moments_range_lockstep,
// stft_wave.h; table: build_mel_moments) — every wave of the workgroup takes a share of the groups (mel_mom_splits: shares of
// equal cost), from the top down, and walks the one group above its share along with it instead of waiting for the wave that
// owns it: one more barrier per frame.
template <int LOG2_NC, int OUT, bool REUSE, int VT>
__global__ __launch_bounds__(BlockFft<LOG2_NC>::T / VT) __attribute__((amdgpu_waves_per_eu(VT == 2 ? 2 : 1))) void stft_block_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_tiles,
    const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw, float *__restrict__ minmax, const uint32_t *__restrict__ mel_tab,
    uint32_t mel_groups, uint32_t n_mel) {
    using B = BlockFft<LOG2_NC>;
    constexpr bool AMP = OUT == 1, MELF = OUT == 2;
    static_assert(!MELF || VT == 1, "mel rows: one virtual thread per thread");
    constexpr int T = B::T, NC = B::NC, TT = T / VT;
    static_assert(VT == 1 || VT == 2, "one or two virtual threads");
    static_assert(VT == 1 || (TT % B::NS_B == 0 && (!B::R2_FIRST || TT % B::NS_A == 0)), "shared pass constants");
    constexpr bool WIN_REGS = LOG2_NC == 12 && !REUSE && VT == 1;  // 32 VGPRs: with three sets of pass constants (n_fft 16384) or the resident samples they do not fit
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];  // BUF_LEN cf32 (n_fft 16384, VT 1: 2 x 68 KB) + the (min, max) scratch
    // n_fft 16384 (one workgroup per CU either way): two exchange buffers used alternately, see the frame loop.  n_fft 8192
    // keeps one: a second would cost its full-reload variant the third workgroup per CU (3.2 -> 3.6 ms on 3840/960/8192)
    // and gains nothing at two (0.72 ms either way).
    constexpr bool DBUF = block_double_buffered(LOG2_NC, VT);
    cf32 *const buf = reinterpret_cast<cf32 *>(smem_raw);
    float *const red = reinterpret_cast<float *>(buf + (DBUF ? 2 : 1) * B::BUF_LEN);
    cf32 *wr = buf, *nx = buf + (DBUF ? B::BUF_LEN : 0);
#define TH_BLOCK_SWAP()                                                                        \
    do {                                                                                       \
        if constexpr (DBUF) {                                                                  \
            cf32 *const tmp_ = wr;                                                             \
            wr = nx;                                                                           \
            nx = tmp_;                                                                         \
        } else {                                                                               \
            __syncthreads(); /* one buffer: its reads are done before the next writes */      \
        }                                                                                      \
    } while (0)
#define TH_VT _Pragma("unroll") for (int v = 0; v < VT; v++)
    const uint32_t t = threadIdx.x;  // virtual thread ids: t + TT v
    const FrameCursor cur = cursor_at(g, jobs, chunk_tab, n_tiles, blockIdx.x);
    if (!cur.valid) return;
    // pass constants: in registers for the whole launch — except at n_fft 32768 with 1024 threads (128 VGPRs each): there
    // every pass loads its ten constants (L2-resident table) when it starts.  VT = 2: the constants of the last pass differ
    // between the two virtual threads (sub-size T): two sets, loaded per frame
    // VT = 2: no constants are resident either; a pass's constants are requested right behind the previous pass's LDS stores,
    // where the 64 registers of z are free, and land during the barrier and the exchange reads
    constexpr bool TW_RES = LOG2_NC <= 13 && VT == 1, TWC_RES = TW_RES;
    cf32 wA[B::NTW], wB[B::NTW], wC[VT][B::NTW];
    if constexpr (TW_RES) {
        if constexpr (B::R2_FIRST) B::template load_tw<B::NS_A>(t, wA, tw);
        B::template load_tw<B::NS_B>(t, wB, tw);
        B::template load_tw<B::NS_C>(t, wC[0], tw);
    }
    cf32 stw_t[VT];
    TH_VT stw_t[v] = tw[t + (uint32_t)TT * v];
    cf32 rw[WIN_REGS ? 16 : 1];
    if constexpr (WIN_REGS) {
#pragma unroll
        for (int m = 0; m < 16; m++) rw[m] = wtab_g[t + (uint32_t)T * m];
    }
    // (!TW_RES: an opaque copy of the thread index per use, or the loads — loop-invariant — are hoisted out of the frame loop
    // and 60 registers stay live across it: 71 spilled VGPRs at n_fft 32768)
    auto tw_lane = [&](uint32_t tv) {
        uint32_t x_ = tv;
        asm volatile("" : "+v"(x_));
        return x_;
    };
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    cf32 x[VT][16];  // raw samples of the frame
    auto fetch = [&](uint32_t f) {
        // the frame's n_fft-sample span starts at e0 (interior frames only: the whole span is inside the channel)
        const int64_t e0 = (int64_t)f * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
        TH_VT {
#pragma unroll
            for (int m = 0; m < 16; m++) {
                const gptr<const float> p = cur.wav + (e0 + 2 * (int64_t)(t + (uint32_t)TT * v + (uint32_t)T * m));
                x[v][m] = {p[0], p[1]};
            }
        }
    };
    // MELF: this wave's share of the filterbank's groups, the same for every frame (ranges of equal cost from the table, mel_mom_splits)
    uint32_t g_lo = 0, g_hi = 0;
    if constexpr (MELF) {
        constexpr uint32_t n_wv = (uint32_t)(TT / 64);
        static_assert(n_wv == 4 || n_wv == 8, "the table carries the ranges of 4 and of 8 waves");
        const uint32_t wv = __builtin_amdgcn_readfirstlane(t >> 6);
        typedef const __attribute__((address_space(4))) uint32_t *cptr32;
        const cptr32 tb = (cptr32)(uintptr_t)mel_tab;
        if (tb[3] != 0) {
            const uint32_t b0 = (n_wv == 8 ? MEL_MOM_SPLIT8_BYTE : MEL_MOM_SPLIT4_BYTE) + wv, b1 = b0 + 1u;
            g_lo = (tb[b0 >> 2] >> (8u * (b0 & 3u))) & 255u;
            g_hi = (tb[b1 >> 2] >> (8u * (b1 & 3u))) & 255u;
        } else {  // (more than 255 groups: equal counts)
            const uint32_t per = (mel_groups + n_wv - 1u) / n_wv;
            g_lo = min(wv * per, mel_groups);
            g_hi = min(g_lo + per, mel_groups);
        }
    }
#if defined(TH_BLK_MEL_PROF)  // development build (scripts/build_variant.sh prof -DTH_BLK_MEL_PROF): shader clocks of the mel epilogue's parts, printed by two workgroups
    uint64_t prof_acc[5] = {0, 0, 0, 0, 0}, prof_q[4] = {0, 0, 0, 0}, prof_last = __builtin_readcyclecounter();
#endif
    if constexpr (REUSE) fetch(cur.f);
    for (uint32_t f = cur.f; f < cur.f1; f++) {
        if constexpr (!REUSE) fetch(f);
        cf32 z[VT][16];
        TH_VT {
            const uint32_t tl = (TW_RES && VT == 1) ? t : tw_lane(t + (uint32_t)TT * v);  // (keeps the 16 window pairs from being hoisted out of the loop too)
#pragma unroll
            for (int m = 0; m < 16; m++) {
                cf32 w;
                if constexpr (WIN_REGS) w = rw[m];
                else w = wtab_g[tl + (uint32_t)T * m];
                z[v][m] = {x[v][m].re * w.re, x[v][m].im * w.im};
            }
        }
        if constexpr (REUSE) {  // the next frame: four slots down, the new hop requested now
            const uint32_t fn = f + 1 < cur.f1 ? f + 1 : f;  // (the last frame of a chunk re-reads its own hop: in bounds, never used)
            const int64_t e0n = (int64_t)fn * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
            TH_VT {
#pragma unroll
                for (int m = 0; m < 12; m++) x[v][m] = x[v][m + 4];
#pragma unroll
                for (int m = 12; m < 16; m++) {
                    const gptr<const float> p = cur.wav + (e0n + 2 * (int64_t)(t + (uint32_t)TT * v + (uint32_t)T * m));
                    x[v][m] = {p[0], p[1]};
                }
            }
        }
        // DBUF: exchange n goes through buffer n mod 2: a thread that has passed the barrier of exchange n - 1 knows that
        // every thread is done reading exchange n - 2, so ONE barrier per exchange (between its writes and its reads) is
        // enough; with a single buffer TH_BLOCK_SWAP is the second barrier, before the next writes (n_fft 16384: 4 barriers
        // per frame instead of 8, 1.12 -> 1.01 ms).
        constexpr bool EARLY = VT == 2;  // (see TW_RES)
        TH_VT B::pass_first(t + (uint32_t)TT * v, z[v], wr);
        if constexpr (EARLY) {
            if constexpr (B::R2_FIRST) B::template load_tw<B::NS_A>(tw_lane(t), wA, tw);
            else B::template load_tw<B::NS_B>(tw_lane(t), wB, tw);
        }
        __syncthreads();
        TH_VT B::template read_in<B::FIRST_LAYOUT>(t + (uint32_t)TT * v, z[v], wr);
        TH_BLOCK_SWAP();
        if constexpr (B::R2_FIRST) {
            if constexpr (!TW_RES && !EARLY) B::template load_tw<B::NS_A>(tw_lane(t), wA, tw);
            TH_VT B::template pass_mid_compute<B::NS_A>(z[v], wA);
            TH_VT B::template pass_mid_store<B::NS_A>(t + (uint32_t)TT * v, z[v], wr);
            if constexpr (EARLY) B::template load_tw<B::NS_B>(tw_lane(t), wB, tw);
            __syncthreads();
            TH_VT B::template read_in<B::NS_A>(t + (uint32_t)TT * v, z[v], wr);
            TH_BLOCK_SWAP();
        }
        if constexpr (!TW_RES && !EARLY) B::template load_tw<B::NS_B>(tw_lane(t), wB, tw);
        TH_VT B::template pass_mid_compute<B::NS_B>(z[v], wB);
        TH_VT B::template pass_mid_store<B::NS_B>(t + (uint32_t)TT * v, z[v], wr);
        if constexpr (EARLY) TH_VT B::template load_tw<B::NS_C>(tw_lane(t + (uint32_t)TT * v), wC[v], tw);
        __syncthreads();
        TH_VT B::template read_in<B::NS_B>(t + (uint32_t)TT * v, z[v], wr);
        TH_BLOCK_SWAP();
        if constexpr (!TWC_RES && !EARLY) B::template load_tw<B::NS_C>(tw_lane(t), wC[0], tw);
        TH_VT B::pass_last(z[v], wC[v]);
        TH_VT B::write_z(t + (uint32_t)TT * v, z[v], wr);
        __syncthreads();
        const gptr<float> row = cur.spec + (size_t)f * cur.spec_pitch;
        float *const amp_f = reinterpret_cast<float *>(DBUF ? nx : wr);  // MELF: the buffer the split pass writes its amplitudes to (wr behind TH_BLOCK_SWAP)
        auto emit = [&](uint32_t k, float p) {
            if constexpr (MELF) {
                amp_f[k] = power_to_amp_scaled(p);
            } else if constexpr (AMP) {  // amplitude rows for the matrix-core mel path
                TH_BLOCK_STORE(&row[k], power_to_amp(p));
            } else {
                const float d = power_to_dB(p);
                TH_BLOCK_STORE(&row[k], d);
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        };
        if constexpr (VT == 1) {
            cf32 zm[8];
            B::split_read(t, wr, zm);
            TH_BLOCK_SWAP();
            B::split_compute(t, z[0], zm, stw_t[0], emit);
        } else {  // one virtual thread after the other (16 registers of mirror partners instead of 32), the barrier behind both
            TH_VT {
                cf32 zm[8];
                B::split_read(t + (uint32_t)TT * v, wr, zm);
                B::split_compute(t + (uint32_t)TT * v, z[v], zm, stw_t[v], emit);
            }
            TH_BLOCK_SWAP();
        }
        if constexpr (MELF) {
#if defined(TH_BLK_MEL_PROF)
            const uint64_t pA = __builtin_readcyclecounter();
#endif
            __syncthreads();  // the amplitude row is complete
#if defined(TH_BLK_MEL_PROF)
            const uint64_t pB = __builtin_readcyclecounter();
#endif
            if (g_lo < g_hi) {  // wave-uniform
                mel_moments_range_lockstep(t & 63u, amp_f, as_global(mel_tab), g_lo, g_hi, min(g_hi + 1u, mel_groups), [&](uint32_t m, float v) {
                    if (m < n_mel) {
                        const float d = amp_to_dB_fast(v);
                        TH_BLOCK_STORE(&row[m], d);
                        lmin = nmin(lmin, d);
                        lmax = nmax(lmax, d);
                    }
                }
#if defined(TH_BLK_MEL_PROF)
                , prof_q
#endif
                );
            }
#if defined(TH_BLK_MEL_PROF)
            const uint64_t pC = __builtin_readcyclecounter();
#endif
            TH_BLOCK_SWAP();  // (the amplitude row's buffer is the next frame's first exchange buffer: two buffers — the other one; one — a barrier)
#if defined(TH_BLK_MEL_PROF)
            {
                const uint64_t pD = __builtin_readcyclecounter();
                prof_acc[0] += pB - pA;
                prof_acc[1] += pC - pB;
                prof_acc[2] += pD - pC;
                prof_acc[3] += pD - prof_last;
                prof_last = pD;
                prof_acc[4] += 1;
            }
#endif
        }
        {   // complete the row's last 128-byte line (see wave_frame)
            const uint32_t height = MELF ? n_mel : (uint32_t)(NC + 1), padn = cur.spec_pitch - height;
            if (t - 1u < ((padn < 32u && cur.spec_pitch % 32u == 0) ? padn : 0u)) row[height - 1u + t] = 0.0f;
        }
    }
#if defined(TH_BLK_MEL_PROF)
    if constexpr (MELF)
        if ((t & 63u) == 0 && (blockIdx.x == 3 || blockIdx.x == 1000))
            printf("blk %u wave %u share [%u, %u): frames %llu  barrier %llu  epilogue %llu  swap %llu  frame %llu cycles; per frame: batches %llu load-wait %llu walk %llu emit %llu\n", blockIdx.x, t >> 6, g_lo, g_hi,
                   (unsigned long long)prof_acc[4], (unsigned long long)(prof_acc[0] / prof_acc[4]), (unsigned long long)(prof_acc[1] / prof_acc[4]),
                   (unsigned long long)(prof_acc[2] / prof_acc[4]), (unsigned long long)(prof_acc[3] / prof_acc[4]), (unsigned long long)(prof_q[3] / prof_acc[4]),
                   (unsigned long long)(prof_q[0] / prof_acc[4]), (unsigned long long)(prof_q[1] / prof_acc[4]), (unsigned long long)(prof_q[2] / prof_acc[4]));
#endif
    if (minmax != nullptr) {  // one (min, max) pair per chunk, folded per channel by wave_post_kernel
        lmin = wave_min(lmin);
        lmax = wave_max(lmax);
        if ((t & 63u) == 0) {
            red[2 * (t >> 6)] = lmin;
            red[2 * (t >> 6) + 1] = lmax;
        }
        __syncthreads();
        if (t == 0) {
            float a = red[0], b = red[1];
            for (int w = 1; w < TT / 64; w++) {
                a = nmin(a, red[2 * w]);
                b = nmax(b, red[2 * w + 1]);
            }
            minmax[2 * (size_t)cur.t] = a;
            minmax[2 * (size_t)cur.t + 1] = b;
        }
    }
}

#undef TH_BLOCK_SWAP

// ------------------------------------------------------------------------------------------
// n_fft 65536 (round 5; VERDICT r4 #7: the generic kernel through global scratch ran at 0.017 of the roofline): the
// workgroup-per-frame plan with PLANAR exchanges (BlockFft<15>: Nc = 32768 = 8 x 16 x 16 x 16).  The 256 KB exchange image does
// not fit the CU's LDS as complex slots, its real parts do (139 KB): every exchange runs twice — real parts out, barrier,
// real parts in, barrier, imaginary parts out, barrier, imaginary parts in, barrier — 512 threads, VT = 4 virtual threads each
// (64 complex points per thread in 128 VGPRs; a pass's constants are loaded when it starts: one set per pass, the last
// pass's per virtual thread).  One workgroup per CU; the frame's samples and window pairs come from L2 / HBM once per frame
// (hop = n_fft / 4: 256 KB of audio re-read + 256 KB of window per 64 KB of new samples — the L2 serves most of it).
// ------------------------------------------------------------------------------------------
template <int LOG2_NC, bool AMP, int VT>
__global__ __launch_bounds__(BlockFft<LOG2_NC>::T / VT) void stft_block_planar_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_tiles,
    const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw, float *__restrict__ minmax) {
    using B = BlockFft<LOG2_NC>;
    constexpr int T = B::T, NC = B::NC, TT = T / VT;
    static_assert(B::PLANAR && B::R2_FIRST && TT % B::NS_B == 0 && TT % B::NS_A == 0, "planar plan, shared constants of the first two twiddled passes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];  // BUF_LEN floats + the (min, max) scratch
    cf32 *const buf = reinterpret_cast<cf32 *>(smem_raw);
    float *const red = reinterpret_cast<float *>(smem_raw) + B::BUF_LEN;
    const uint32_t t = threadIdx.x;  // virtual thread ids: t + TT v
    const FrameCursor cur = cursor_at(g, jobs, chunk_tab, n_tiles, blockIdx.x);
    if (!cur.valid) return;
    auto tw_lane = [&](uint32_t tv) {  // (opaque per use: the loop-invariant constant loads must not be hoisted out of the frame loop)
        uint32_t x_ = tv;
        asm volatile("" : "+v"(x_));
        return x_;
    };
    // (the split twiddle W^t is loaded per frame, through the opaque thread id: resident, its sixteen products with the W^(T c)
    // per virtual thread are loop-invariant — 64 registers hoisted out of the frame loop and spilled)
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    // one exchange: STORE and READ are called with the part as template argument
#define TH_PLANAR_EXCHANGE(STORE0, READ0, STORE1, READ1) \
    do {                                                 \
        TH_VT { STORE0; TH_SCHED_BARRIER(); }            \
        __syncthreads();                                 \
        TH_VT { READ0; }                                 \
        __syncthreads();                                 \
        TH_VT { STORE1; TH_SCHED_BARRIER(); }            \
        __syncthreads();                                 \
        TH_VT { READ1; }                                 \
        __syncthreads();                                 \
    } while (0)
    for (uint32_t f = cur.f; f < cur.f1; f++) {
        // (an opaque copy of the virtual thread ids per frame: every LDS / row address below is "f(id) + immediate" and
        // loop-invariant — hoisted out of the frame loop they are ~200 live registers, i.e. 640 bytes of scratch per lane)
        uint32_t tv[VT];
        TH_VT tv[v] = tw_lane(t + (uint32_t)TT * v);
        cf32 z[VT][16];
        {   // the frame's n_fft-sample span starts at e0 (interior frames only: the whole span is inside the channel)
            const int64_t e0 = (int64_t)f * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
            TH_VT {
                // (one virtual thread's 32 loads at a time: left to itself the scheduler requests all 128 first and spills)
                const uint32_t tl = tv[v];
#pragma unroll
                for (int m = 0; m < 16; m++) {
                    const gptr<const float> p = cur.wav + (e0 + 2 * (int64_t)(tl + (uint32_t)T * m));
                    const cf32 w = wtab_g[tl + (uint32_t)T * m];
                    z[v][m] = {p[0] * w.re, p[1] * w.im};
                }
                TH_SCHED_BARRIER();
            }
        }
        TH_VT {
            B::pass_first_compute(z[v]);
            TH_SCHED_BARRIER();
        }
        TH_PLANAR_EXCHANGE(B::template pass_first_store<0>(tv[v], z[v], buf),
                           (B::template read_in<B::FIRST_LAYOUT, 0>(tv[v], z[v], buf)),
                           B::template pass_first_store<1>(tv[v], z[v], buf),
                           (B::template read_in<B::FIRST_LAYOUT, 1>(tv[v], z[v], buf)));
        {
            cf32 wA[B::NTW];
            B::template load_tw<B::NS_A>(tv[0], wA, tw);
            TH_VT {
                B::template pass_mid_compute<B::NS_A>(z[v], wA);
                TH_SCHED_BARRIER();  // (one virtual thread after the other: interleaved, their temporaries add up)
            }
        }
        TH_PLANAR_EXCHANGE((B::template pass_mid_store<B::NS_A, 0>(tv[v], z[v], buf)),
                           (B::template read_in<B::NS_A, 0>(tv[v], z[v], buf)),
                           (B::template pass_mid_store<B::NS_A, 1>(tv[v], z[v], buf)),
                           (B::template read_in<B::NS_A, 1>(tv[v], z[v], buf)));
        {
            cf32 wB[B::NTW];
            B::template load_tw<B::NS_B>(tv[0], wB, tw);
            TH_VT {
                B::template pass_mid_compute<B::NS_B>(z[v], wB);
                TH_SCHED_BARRIER();
            }
        }
        TH_PLANAR_EXCHANGE((B::template pass_mid_store<B::NS_B, 0>(tv[v], z[v], buf)),
                           (B::template read_in<B::NS_B, 0>(tv[v], z[v], buf)),
                           (B::template pass_mid_store<B::NS_B, 1>(tv[v], z[v], buf)),
                           (B::template read_in<B::NS_B, 1>(tv[v], z[v], buf)));
        TH_VT {
            cf32 wC[B::NTW];
            B::template load_tw<B::NS_C>(tv[v], wC, tw);
            B::pass_last(z[v], wC);
            TH_SCHED_BARRIER();
        }
        cf32 zm[VT][8];
        TH_PLANAR_EXCHANGE(B::template write_z<0>(tv[v], z[v], buf), B::template split_read<0>(tv[v], buf, zm[v]),
                           B::template write_z<1>(tv[v], z[v], buf), B::template split_read<1>(tv[v], buf, zm[v]));
        const gptr<float> row = cur.spec + (size_t)f * cur.spec_pitch;
        auto emit = [&](uint32_t k, float p) {
            if constexpr (AMP) {  // amplitude rows for the two-kernel mel path
                TH_BLOCK_STORE(&row[k], power_to_amp(p));
            } else {
                const float d = power_to_dB(p);
                TH_BLOCK_STORE(&row[k], d);
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        };
        TH_VT {
            B::split_compute(tv[v], z[v], zm[v], tw[tv[v]], emit);
            TH_SCHED_BARRIER();
        }
        {   // complete the row's last 128-byte line (see wave_frame)
            const uint32_t height = (uint32_t)(NC + 1), padn = cur.spec_pitch - height;
            if (t - 1u < ((padn < 32u && cur.spec_pitch % 32u == 0) ? padn : 0u)) row[height - 1u + t] = 0.0f;
        }
    }
#undef TH_PLANAR_EXCHANGE
    if (minmax != nullptr) {  // one (min, max) pair per chunk, folded per channel by wave_post_kernel
        lmin = wave_min(lmin);
        lmax = wave_max(lmax);
        if ((t & 63u) == 0) {
            red[2 * (t >> 6)] = lmin;
            red[2 * (t >> 6) + 1] = lmax;
        }
        __syncthreads();
        if (t == 0) {
            float a = red[0], b = red[1];
            for (int w = 1; w < TT / 64; w++) {
                a = nmin(a, red[2 * w]);
                b = nmax(b, red[2 * w + 1]);
            }
            minmax[2 * (size_t)cur.t] = a;
            minmax[2 * (size_t)cur.t + 1] = b;
        }
    }
}

#undef TH_VT

#if !defined(TH_BLOCK_VT_13)
#define TH_BLOCK_VT_13 1  // virtual threads per thread of the n_fft 16384 block kernel: 2 measured 1.35 ms against 1.01 (profiles/r04_ab_block_virtual_threads.txt)
#endif
#if !defined(TH_BLOCK_VT_14)
#define TH_BLOCK_VT_14 2  // ... of the n_fft 32768 kernel: 1.94-1.98 ms against 2.19 (hop = n_fft / 4), 3.26 against 3.69 (19200 / 4800)
#endif
#if !defined(TH_BLOCK_VT2_REUSE)
#define TH_BLOCK_VT2_REUSE 0  // resident samples under VT = 2: 1.98 ms against 1.94 without at n_fft 32768 (256 VGPRs + 120 bytes of scratch)
#endif
template <int LOG2_NC, int OUT, bool REUSE, int VT>
static hipError_t launch_block_t(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_chunk_tab, uint32_t n_tiles,
                               const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, hipStream_t s, const WaveOut *mel = nullptr) {
    using B = BlockFft<LOG2_NC>;
    auto kern = stft_block_kernel<LOG2_NC, OUT, REUSE, VT>;
    constexpr size_t lds = sizeof(cf32) * (block_double_buffered(LOG2_NC, VT) ? 2 : 1) * B::BUF_LEN + sizeof(float) * 2 * (B::T / VT / 64);
    static_assert(lds + 64 <= 160 * 1024, "the exchange buffers fit the CU's LDS");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(B::T / VT), lds, s, g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_minmax,
                       mel ? mel->mel_tab : nullptr, mel ? mel->mel_groups : 0u, mel ? mel->n_mel : 0u);
    return hipGetLastError();
}

template <int LOG2_NC, int OUT>
static hipError_t launch_block(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_chunk_tab, uint32_t n_tiles,
                               const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, hipStream_t s, const WaveOut *mel = nullptr) {
    constexpr int VT = LOG2_NC == 13 ? TH_BLOCK_VT_13 : LOG2_NC == 14 ? TH_BLOCK_VT_14 : 1;
    static_assert(OUT != 2 || VT == 1, "mel rows: sizes whose block kernel runs one virtual thread per thread");
    // (every frame of a chunk then sits exactly four slots behind its predecessor; at n_fft 32768 only with two virtual threads:
    // 1024 threads have 128 VGPRs each)
    if constexpr (LOG2_NC <= 13 || VT == 2)
        if (g.hop * 4 == g.n_fft && (VT == 1 || TH_BLOCK_VT2_REUSE)) return launch_block_t<LOG2_NC, OUT, true, VT>(g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_minmax, s, mel);
    return launch_block_t<LOG2_NC, OUT, false, VT>(g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_minmax, s, mel);
}


#if !defined(TH_BLOCK_VT_15)
#define TH_BLOCK_VT_15 4  // virtual threads per thread of the planar n_fft 65536 kernel (512 threads, 256 VGPRs)
#endif
template <int LOG2_NC, bool AMP>
static hipError_t launch_block_planar(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_chunk_tab, uint32_t n_tiles,
                                      const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, hipStream_t s) {
    using B = BlockFft<LOG2_NC>;
    constexpr int VT = TH_BLOCK_VT_15;
    auto kern = stft_block_planar_kernel<LOG2_NC, AMP, VT>;
    constexpr size_t lds = sizeof(float) * B::BUF_LEN + sizeof(float) * 2 * (B::T / VT / 64);
    static_assert(lds + 64 <= 160 * 1024, "one part of the exchange image fits the CU's LDS");
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_tiles), dim3(B::T / VT), lds, s, g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_minmax);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_minmax_init(float *d_minmax, uint32_t n_chan, hipStream_t s) {
    if (!d_minmax || !n_chan) return hipSuccess;
    hipLaunchKernelGGL(minmax_init_kernel, dim3((n_chan + 255) / 256), dim3(256), 0, s, d_minmax, n_chan);
    return hipGetLastError();
}

// find_min_max over every resident spec (core/mod.rs:169-178): one block reduces the per-channel (min, max) pairs
// the STFT launch left in d_minmax to out = [min, -max] (so that ONE element-wise MIN all-reduce merges ranks).
__global__ __launch_bounds__(256) void minmax_reduce_kernel(const float *__restrict__ minmax, uint32_t n_chan,
                                                            float *__restrict__ out, float dB_range,
                                                            float *__restrict__ out_range) {
    __shared__ float smn[4], smx[4];
    float mn = __builtin_inff(), mx = -__builtin_inff();
    for (uint32_t i = threadIdx.x; i < n_chan; i += 256) {
        mn = nmin(mn, minmax[2 * i]);
        mx = nmax(mx, minmax[2 * i + 1]);
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((threadIdx.x & 63u) == 0) {
        smn[threadIdx.x >> 6] = mn;
        smx[threadIdx.x >> 6] = mx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mn = nmin(nmin(smn[0], smn[1]), nmin(smn[2], smn[3]));
        mx = nmax(nmax(smx[0], smx[1]), nmax(smx[2], smx[3]));
        if (out != nullptr) {
            out[0] = mn;
            out[1] = -mx;
        }
        if (out_range != nullptr) {  // update_spec_imgs' clamp (core/mod.rs:179-180), as db_range_kernel (kernels_image.hip)
            mx = fminf(mx, 0.0f);
            out_range[0] = fmaxf(mn, mx - dB_range);
            out_range[1] = mx;
        }
    }
}

hipError_t launch_minmax_reduce(const float *d_minmax, uint32_t n_chan, float *d_out, float dB_range, float *d_range, hipStream_t s) {
    hipLaunchKernelGGL(minmax_reduce_kernel, dim3(1), dim3(256), 0, s, d_minmax, n_chan, d_out, dB_range, d_range);
    return hipGetLastError();
}

// After the wave kernel: one block per channel folds the channel's per-chunk (min, max) pairs (its jobs' tiles are
// one contiguous range) into the channel's slot — a plain store when every frame of the channel was in the wave launch
// (then nobody has to initialise the slot), atomics when other kernels contribute too — and block 0 rewinds the chunk
// queue for the next launch.  With chunk_mm == NULL only the queue is rewound.
__global__ __launch_bounds__(256) void wave_post_kernel(const WavePostJob *__restrict__ pj, const float *__restrict__ chunk_mm,
                                                        float *__restrict__ mm_slots, int store, uint32_t *__restrict__ queue_head,
                                                        float dB_range, float *__restrict__ d_range) {
    __shared__ float red[8];
    const uint32_t tid = threadIdx.x;
    if (blockIdx.x == 0 && tid == 0) *queue_head = 0;
    if (chunk_mm == nullptr) return;
    const WavePostJob job = pj[blockIdx.x];
    const gptr<const float2> mm = reinterpret_cast<gptr<const float2>>(as_global(chunk_mm));
    float mn = __builtin_inff(), mx = -__builtin_inff();
    // four independent loads in flight per thread: a long single-track channel (thousands of chunks) is latency-bound
    for (uint32_t t = job.t0 + tid; t < job.t1; t += 4 * 256) {
        float2 v[4];
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) v[u] = mm[min(t + 256u * u, job.t1 - 1u)];  // clamped repeats do not change min / max
#pragma unroll
        for (uint32_t u = 0; u < 4; u++) {
            mn = nmin(mn, v[u].x);
            mx = nmax(mx, v[u].y);
        }
    }
    mn = wave_min(mn);
    mx = wave_max(mx);
    if ((tid & 63u) == 0) {
        red[2 * (tid >> 6)] = mn;
        red[2 * (tid >> 6) + 1] = mx;
    }
    __syncthreads();
    if (tid == 0) {
        mn = nmin(nmin(red[0], red[2]), nmin(red[4], red[6]));
        mx = nmax(nmax(red[1], red[3]), nmax(red[5], red[7]));
        if (store) {
            mm_slots[2 * job.mm_index] = mn;
            mm_slots[2 * job.mm_index + 1] = mx;
            if (d_range != nullptr) {  // single-channel batch: the global range is this channel's (core/mod.rs:179-180)
                const float hi = fminf(mx, 0.0f);
                d_range[0] = fmaxf(mn, hi - dB_range);
                d_range[1] = hi;
            }
        } else {
            atomic_min_f32(&mm_slots[2 * job.mm_index], mn);
            atomic_max_f32(&mm_slots[2 * job.mm_index + 1], mx);
        }
    }
}
hipError_t launch_wave_post(const WavePostJob *d_pj, uint32_t n_pj, const float *d_chunk_mm, float *d_mm_slots, bool store,
                            uint32_t *d_queue_head, float dB_range, float *d_range, hipStream_t s) {
    const bool fold = d_chunk_mm != nullptr && n_pj != 0;
    hipLaunchKernelGGL(wave_post_kernel, dim3(fold ? n_pj : 1), dim3(256), 0, s, d_pj, fold ? d_chunk_mm : nullptr, d_mm_slots,
                       store ? 1 : 0, d_queue_head, dB_range, (fold && store && n_pj == 1) ? d_range : nullptr);
    return hipGetLastError();
}

size_t stft_generic_lds_bytes(const StftGeom &g) { return 2 * (size_t)g.nc * sizeof(cf32) * stft_generic_frames_par(g); }
// n_fft > 16384: the frame buffers do not fit LDS; workgroups and bytes of global scratch the launch needs (0: LDS variant)
uint32_t stft_generic_scratch_grid(const StftGeom &g, uint32_t n_tiles, uint32_t n_cu) {
    if (stft_generic_lds_bytes(g) <= 128 * 1024) return 0;
    const size_t per_wg = stft_generic_lds_bytes(g);
    const size_t cap = ((size_t)1 << 28) / per_wg;  // at most 256 MiB of scratch per plan (1024 workgroups at n_fft 32768)
    const size_t want = (size_t)n_cu * 4;           // 4 workgroups of 256 threads per CU
    return (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)n_tiles, want, cap}));
}
size_t stft_generic_scratch_bytes(const StftGeom &g, uint32_t n_tiles, uint32_t n_cu) {
    return (size_t)stft_generic_scratch_grid(g, n_tiles, n_cu) * stft_generic_lds_bytes(g);
}

// Bluestein plans: M = 2^m >= 2 Nc - 1; two M-point cf64 buffers per workgroup, at most 1 GiB of scratch per plan
uint32_t stft_bluestein_m(const StftGeom &g) {
    uint32_t M = 1;
    while (M < 2u * g.nc - 1u) M <<= 1;
    return M;
}
static uint32_t stft_bluestein_grid(const StftGeom &g, uint32_t n_tiles, uint32_t n_cu) {
    const size_t per_wg = 2 * (size_t)stft_bluestein_m(g) * sizeof(cf64);
    return (uint32_t)std::max<size_t>(1, std::min<size_t>({(size_t)n_tiles, (size_t)n_cu * 4, ((size_t)1 << 30) / per_wg}));
}
size_t stft_bluestein_scratch_bytes(const StftGeom &g, uint32_t n_tiles, uint32_t n_cu) {
    return n_tiles ? (size_t)stft_bluestein_grid(g, n_tiles, n_cu) * 2 * (size_t)stft_bluestein_m(g) * sizeof(cf64) : 0;
}
hipError_t launch_stft_bluestein(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan, uint32_t n_tiles,
                                 const float *d_window, const void *d_chirp, const void *d_bhat, const void *d_twm, const void *d_tws,
                                 const float *d_mel_fb, const uint32_t *d_mel_lo, const uint32_t *d_mel_hi, float *d_minmax, hipStream_t s,
                                 void *d_scratch, uint32_t n_cu) {
    if (!n_tiles) return hipSuccess;
    if (d_scratch == nullptr || d_chirp == nullptr || d_bhat == nullptr || d_twm == nullptr || d_tws == nullptr) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stft_bluestein_kernel, dim3(stft_bluestein_grid(g, n_tiles, n_cu)), dim3(GEN_THREADS), 0, s, g, d_jobs, d_tile_start, n_chan,
                       n_tiles, d_window, static_cast<const cf64 *>(d_chirp), static_cast<const cf64 *>(d_bhat), static_cast<const cf64 *>(d_twm),
                       static_cast<const cf64 *>(d_tws), d_mel_fb, d_mel_lo, d_mel_hi, d_minmax, static_cast<cf64 *>(d_scratch), stft_bluestein_m(g));
    return hipGetLastError();
}

hipError_t launch_stft_generic(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                               uint32_t n_chan, uint32_t n_tiles, const float *d_window, const cf32 *d_tw,
                               const float *d_mel_fb, const uint32_t *d_mel_lo, const uint32_t *d_mel_hi,
                               float *d_minmax, hipStream_t s, void *d_scratch, uint32_t n_cu) {
    if (!n_tiles) return hipSuccess;
    const uint32_t sgrid = stft_generic_scratch_grid(g, n_tiles, n_cu);
    if (sgrid) {
        if (d_scratch == nullptr) return hipErrorInvalidValue;
        hipLaunchKernelGGL(stft_generic_kernel<true>, dim3(sgrid), dim3(GEN_THREADS), 0, s, g, d_jobs, d_tile_start, n_chan, n_tiles,
                           d_window, d_tw, d_mel_fb, d_mel_lo, d_mel_hi, d_minmax, static_cast<cf32 *>(d_scratch));
        return hipGetLastError();
    }
    const size_t lds = stft_generic_lds_bytes(g);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_generic_kernel<false>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(stft_generic_kernel<false>, dim3(n_tiles), dim3(GEN_THREADS), lds, s, g, d_jobs, d_tile_start,
                       n_chan, n_tiles, d_window, d_tw, d_mel_fb, d_mel_lo, d_mel_hi, d_minmax, (cf32 *)nullptr);
    return hipGetLastError();
}
}  // namespace th

namespace th {

// n_fft 512 (multi-frame kernel): linear dB only
bool stft_wave_supported(const StftGeom &g) {
    // n_fft 512 (multi-frame kernel) ... 4096 (one frame per wave), 8192 / 16384 / 32768 (block kernel); mel plans: fused
    // epilogue at n_fft 512 / 1024 / 2048 where the tables fit (at most 512 mels), else amplitude rows + a second kernel
    // (banded sums or the matrix cores: any mel count)
    return g.odd_m1 == 0 && g.log2_nc >= 8 && g.log2_nc <= 15;  // (15: n_fft 65536, the planar block plan of round 5; powers of two only)
}
bool stft_is_block_plan(const StftGeom &g) { return g.log2_nc >= 12; }
// mel rows in the block kernel's own epilogue (moment form): n_fft 8192 / 16384 where stft_block_kernel is what runs
// (long_plan: 0 the size's default plan, 1 block kernel, 2 subwave plan — WaveOut::long_plan)
bool stft_block_mel_fused_applies(const StftGeom &g, int long_plan) {
    if (g.odd_m1 != 0 || g.phased != 0 || (g.log2_nc != 12 && g.log2_nc != 13)) return false;
    return long_plan == 1 || (long_plan == 0 && !stft_subwave_default(g));
}
// amplitude floats a lane of that epilogue may address (the exchange buffer), and the most groups it takes
uint32_t stft_block_mel_max_index(const StftGeom &g) { return 2u * (g.nc + g.nc / 16u + 2u) - 64u; }
bool stft_wave_multi_applies(const StftGeom &g, int out_mode) {
    return g.phased == 0 && ((out_mode == 0 && (g.log2_nc == 8 || g.log2_nc == 9)) || ((out_mode == 1 || out_mode == 2) && g.log2_nc == 8));
}

// Staged loads of the multi-frame kernel (launch_wave_multi_t: n_fft 512, even hop, span within 1280 samples) fetch 16-byte
// groups from the iteration's first sample on and clamp a group's start to n_samples - 4.  A group that holds needed samples
// must therefore start at or below n_samples - 4: with hop % 4 == 2 a frame's last two samples sit in a group of their own
// (offset 510 of a frame that starts 2 mod 4 into the grid), so its span has to end 2 samples before the channel does.
// The host keeps frames that end later out of the interior set (they run as boundary frames); 0 for every other plan.
uint32_t stft_wave_multi_tail_guard(const StftGeom &g) {
    const bool staged = g.log2_nc == 8 && g.hop % 2u == 0 && 3u * g.hop + g.n_fft <= 1280u;
    return staged && g.hop % 4u != 0 ? 2u : 0u;
}

#endif  // TH_PART_MAIN
// Waves per workgroup (one persistent workgroup per CU).  Bounded by LDS (tables + one slab per
// wave <= 160 KB) and by the VGPR file (64*WAVES threads => 512/(WAVES/4) VGPRs per lane).
#if !defined(TH_RES12)
#define TH_RES12 1
#endif
#if !defined(TH_WAVES_4096)
#define TH_WAVES_4096 8
#endif
#if !defined(TH_DYN_EVEN_4096)
#define TH_DYN_EVEN_4096 1  // grid-aligned mode without the odd window table where the offsets are always even (0: A/B builds)
#endif
template <int LOG2_NC>
struct WaveLaunchCfg {
    static constexpr int DEFAULT_WAVES = LOG2_NC == 11 ? TH_WAVES_4096 : 12;  // n_fft = 4096: LDS-bound (17 KB slab per wave): 8 waves = 160 KB exactly
    // launch shape of the grid-aligned (phased / dynamic) modes: n_fft 4096 gives one wave's slab to the second window table
    static constexpr int GRID_WAVES = LOG2_NC == 11 ? 7 : DEFAULT_WAVES;
    // register-resident tables by VGPR budget (512 / waves per SIMD); mirror-local path only for bits 2, 3
    static constexpr int resident(int waves) {
        // n_fft 4096: split twiddles in registers (32 VGPRs) free their 16 KB table; with the pass-2 constants too (20
        // more) the eighth wave's slab fits exactly (160 KB)
        // (7 waves: the same, which leaves exactly the 17 KB the second window table of the dynamic mode needs)
        if (LOG2_NC == 11) return waves <= 6 ? 8 : waves <= 8 ? 10 : 0;
        if (LOG2_NC != 10) return 0;
        return waves <= 8 ? 15 : waves <= 12 ? TH_RES12 : 0;
    }
};

// (selector 6 runs the two-frames-per-wave plan of n_fft 1024 at the one-frame plan's default shape: it must be one the
// multi-frame kernel is instantiated for, launch_wave_multi)
static_assert(WaveLaunchCfg<9>::DEFAULT_WAVES == 8 || WaveLaunchCfg<9>::DEFAULT_WAVES == 12 || WaveLaunchCfg<9>::DEFAULT_WAVES == 16,
              "stft_wave_multi_kernel<9> exists for 8, 12 and 16 waves per workgroup");

template <int LOG2_NC, int WAVES, bool PKV = (TH_USE_PK != 0)>
static size_t wave_lds_bytes() {
    using W = WaveFft<LOG2_NC>;
    const bool stw_in_lds = !((WaveLaunchCfg<LOG2_NC>::resident(WAVES) & 8) && W::PAIRED);
    const bool t2_in_lds = !(WaveLaunchCfg<LOG2_NC>::resident(WAVES) & 2);
    const size_t stw_len = (W::PK && PKV) ? (size_t)W::STWP_LEN : (size_t)W::NC;
    return sizeof(cf32) * ((size_t)W::NC + (stw_in_lds ? stw_len : 0) + (t2_in_lds ? W::T2_LEN : 0) + W::T3_LEN + (size_t)WAVES * W::SLAB_LEN);
}

#if !TH_PART_MAIN
template <int LOG2_NC, int WAVES, int SHIFT, int OUT>
static hipError_t launch_wave_t5(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                                 uint32_t n_chan, uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw,
                                 float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, const WaveOut &out, hipStream_t s) {
    // the packed-f32 pipeline (selector 9): instantiated for the headline shape only (n_fft 2048, 12 waves, hop = n_fft / 4, dB rows)
    constexpr bool PK_SHAPE = (TH_AB_VARIANTS != 0) && LOG2_NC == 10 && WAVES == 12 && SHIFT == 4 && OUT == 0;  // (A/B builds only, kernels.h)
    const bool pkv = (TH_USE_PK != 0) || (PK_SHAPE && out.packed != 0);
    auto kern = stft_wave_kernel<LOG2_NC, WAVES, SHIFT, OUT, WaveLaunchCfg<LOG2_NC>::resident(WAVES), (TH_USE_PK != 0)>;
    if constexpr (PK_SHAPE && TH_USE_PK == 0)
        if (pkv) kern = stft_wave_kernel<LOG2_NC, WAVES, SHIFT, OUT, WaveLaunchCfg<LOG2_NC>::resident(WAVES), true>;
    if constexpr (LOG2_NC == 9 && OUT == 3 && WAVES == 12 && TH_USE_PK == 0)  // two workgroups per CU: 80 VGPRs (see the entry point)
        kern = stft_wave_kernel_occ6<LOG2_NC, WAVES, SHIFT, OUT, WaveLaunchCfg<LOG2_NC>::resident(WAVES)>;
    // the sweep chunk schedule (large batches; the host then cut 4-frame chunks): the same shape, scalar pipeline
    bool sweep = false;
    if constexpr (PK_SHAPE) {
        if (out.sweep != 0 && !pkv) {
            if (g.frames_per_tile != 4) return hipErrorInvalidValue;
            kern = stft_wave_kernel<LOG2_NC, WAVES, SHIFT, OUT, WaveLaunchCfg<LOG2_NC>::resident(WAVES), false, true>;
            sweep = true;
        }
    }
    if (out.sweep != 0 && !sweep) return hipErrorInvalidValue;  // (the host only asks for it where it exists)
    const size_t lds = (pkv ? wave_lds_bytes<LOG2_NC, WAVES, true>() : wave_lds_bytes<LOG2_NC, WAVES, false>()) + (sweep ? 128 : 0) + ((OUT == 2 || OUT == 3) && LOG2_NC != 11 ? (size_t)((out.mel_words + 1u) & ~1u) * 4 : 0) +
                       (OUT == 2 && LOG2_NC == 9 && out.mel_slots != 0 ? (size_t)WAVES * MEL_PRF_1024 * sizeof(cf32) : 0) +  // (pieces / gather only)
                       (SHIFT == -1 ? 48 * sizeof(cf32) : 0) +  // phased: zero pairs in front of the window table
                       (SHIFT <= -48 ? 0 : SHIFT <= -16 ? (WaveFft<LOG2_NC>::NC + 128) * sizeof(cf32) : 0);  // dynamic: second table + two prefixes (even offsets only: neither)
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    const uint32_t wg_needed = (n_tiles + WAVES - 1) / WAVES;
    // workgroups per CU: a workgroup holds at most 16 waves; the small n_fft 1024 kernel (69 VGPRs, 4 KB slab per wave) is
    // bound by latency and runs two workgroups per CU when LDS allows
    // (measured, 1024/256 on the bench tracks: 8 / 10 / 12 / 14 / 16 waves in one workgroup 0.78 / 0.71 / 0.66 / 0.64 / 0.63 ms,
    // 2 x 12 waves 0.59, 3 x 12 0.58)
    const uint32_t per_cu = (LOG2_NC == 9 && 2 * lds <= 160 * 1024) ? 2u : 1u;
    const uint32_t grid = wg_needed < n_cu * per_cu ? wg_needed : n_cu * per_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * WAVES), lds, s, g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab,
                       d_tw, d_minmax, d_queue_head, out);
    return hipGetLastError();
}

// amplitude and fused-mel output are only instantiated for the default launch shape of each n_fft (fused mel: n_fft =
// 2048 only — 1024 has no mirror-local last pass, 4096's 7 waves leave no LDS for the table)
template <int LOG2_NC, int WAVES, int SHIFT>
static hipError_t launch_wave_t4(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                                 uint32_t n_chan, uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw,
                                 float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, const WaveOut &out, hipStream_t s) {
    if constexpr (WAVES == ((SHIFT < 0 && SHIFT > -48) ? WaveLaunchCfg<LOG2_NC>::GRID_WAVES : WaveLaunchCfg<LOG2_NC>::DEFAULT_WAVES)) {
        if constexpr (SHIFT >= 0 || (SHIFT <= -16 && LOG2_NC == 11)) {
            if (out.mode == 1)
                return launch_wave_t5<LOG2_NC, WAVES, SHIFT, 1>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab,
                                                                        d_tw, nullptr, d_queue_head, n_cu, out, s);
        }
        // (n_fft 4096, round 6: the moment form, for hop 1024 and the 96 / 88.2 kHz defaults — stft_wave_mel_fits)
        // (round 5: frame pairs — two consecutive frames per pass over the banded table; n_fft 2048, plain or rotating frame loop)
        if constexpr ((LOG2_NC == 10 || LOG2_NC == 9) && SHIFT == 0) {  // (its own instantiation: inlined into OUT = 2 it cost the one-frame epilogue 9 %)
            if (out.mode == 2 && out.mel_moment != 0)
                return launch_wave_t5<LOG2_NC, WAVES, SHIFT, 4>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab,
                                                                        d_tw, d_minmax, d_queue_head, n_cu, out, s);
        }
        if (out.mode == 2 && out.mel_moment != 0 && LOG2_NC <= 10) return hipErrorInvalidValue;  // (wave_shift keeps such launches at SHIFT 0)
        if constexpr (LOG2_NC == 10 || LOG2_NC == 9) {
            if (out.mode == 2 && out.mel_pair != 0) {
                if (out.mel_slots != 0) return hipErrorInvalidValue;  // (banded sums only)
                return launch_wave_t5<LOG2_NC, WAVES, SHIFT, 3>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab,
                                                                        d_tw, d_minmax, d_queue_head, n_cu, out, s);
            }
        }
        // (n_fft 4096: every plain frame loop — hop a multiple of 128 samples with 8, 16 or 4 slots of reuse, or none — and the
        // even-offset grid-aligned shapes of the 96 / 88.2 kHz defaults at t_overlap 4 and 8)
        if constexpr (LOG2_NC <= 10 || (LOG2_NC == 11 && (SHIFT == 8 || SHIFT == 16 || SHIFT == 4 || SHIFT == 0 || SHIFT == -48 - 7 || SHIFT == -48 - 6 || SHIFT == -48 - 3 || SHIFT == -48 - 1 || SHIFT == -48 - 0 || SHIFT == -48 - 13))) {
            if (out.mode == 2)
                return launch_wave_t5<LOG2_NC, WAVES, SHIFT, 2>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab,
                                                                        d_tw, d_minmax, d_queue_head, n_cu, out, s);
        }
    }
    if (out.mode != 0) return hipErrorInvalidValue;
    return launch_wave_t5<LOG2_NC, WAVES, SHIFT, 0>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw,
                                                            d_minmax, d_queue_head, n_cu, out, s);
}

// register-reuse shift of consecutive frames: hop/128 slots when hop is a multiple of 128 samples
// and smaller than n_fft and the window is not zero-padded, else 0 (full reload per frame)
template <int LOG2_NC>
static int wave_shift(const StftGeom &g) {
    constexpr int P = WaveFft<LOG2_NC>::P;
    if (g.hop % 128 != 0) return 0;
    const int sh = (int)(g.hop / 128);
    return (sh >= 1 && sh < P) ? sh : 0;
}

template <int LOG2_NC, int WAVES>
static hipError_t launch_wave_t3(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                                 uint32_t n_chan, uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw,
                                 float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, const WaveOut &out, hipStream_t s) {
    constexpr int P = WaveFft<LOG2_NC>::P;
    if (g.phased == 3) {  // dynamic mode, even offsets only (n_fft 4096: the 96 / 88.2 kHz defaults at eight waves)
        if constexpr (LOG2_NC == 11 && WAVES == WaveLaunchCfg<LOG2_NC>::DEFAULT_WAVES) {
            const uint32_t k = g.hop / 128;
#define TH_DYN_EVEN_CASE(K)                                                                                            \
    if (k == (K))                                                                                                     \
        return launch_wave_t4<LOG2_NC, WAVES, -48 - (K)>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, \
                                                          d_queue_head, n_cu, out, s);
            TH_DYN_EVEN_CASE(7)   // 96 kHz: 3840 / 960
            TH_DYN_EVEN_CASE(6)   // 88.2 kHz: 3528 / 882
            TH_DYN_EVEN_CASE(3)   // 96 kHz, t_overlap 8: 3840 / 480 (3528 / 441 has an odd hop: two tables, seven waves)
            TH_DYN_EVEN_CASE(1)   // t_overlap 16: 3840 / 240, 3528 / 220
            TH_DYN_EVEN_CASE(0)   // t_overlap 32: 3840 / 120, 3528 / 110
            TH_DYN_EVEN_CASE(13)  // 88.2 kHz, t_overlap 2: 3528 / 1764
#undef TH_DYN_EVEN_CASE
        }
        return hipErrorInvalidValue;
    }
    if (g.phased) {  // grid-aligned loads (see stft_wave_kernel): only the default launch shapes are instantiated
        if constexpr (WAVES == WaveLaunchCfg<LOG2_NC>::GRID_WAVES) {
            const uint32_t k = g.hop / 128;
#define TH_DYN_CASE(L2, K)                                                                                            \
    if constexpr (LOG2_NC == (L2))                                                                                    \
        if (g.phased == 2 && k == (K))                                                                                \
            return launch_wave_t4<LOG2_NC, WAVES, -16 - (K)>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw,   \
                                                             d_minmax, d_queue_head, n_cu, out, s);
            if constexpr (LOG2_NC == 10)
                if (g.phased == 1)
                    return launch_wave_t4<LOG2_NC, WAVES, -1>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw,
                                                              d_minmax, d_queue_head, n_cu, out, s);
            TH_DYN_CASE(10, 3)  // 44.1 kHz: 1764 / 441
            TH_DYN_CASE(10, 2)  // 32 kHz: 1280 / 320
            TH_DYN_CASE(10, 7)  // 48 kHz, t_overlap 2: 1920 / 960
            TH_DYN_CASE(10, 6)  // 44.1 kHz, t_overlap 2: 1764 / 882
            TH_DYN_CASE(10, 1)  // 48 kHz, t_overlap 8: 1920 / 240
            TH_DYN_CASE(10, 0)  // 48 kHz, t_overlap 16 / 32: 1920 / 120, 1920 / 60
            TH_DYN_CASE(9, 1)   // 16 kHz: 640 / 160, 22.05 kHz: 884 / 221
            TH_DYN_CASE(9, 0)   // 16 kHz, t_overlap 8 .. 32: 640 / 80
            TH_DYN_CASE(11, 7)  // 96 kHz: 3840 / 960
            TH_DYN_CASE(11, 6)  // 88.2 kHz: 3528 / 882
            TH_DYN_CASE(11, 3)  // 96 / 88.2 kHz, t_overlap 8: 3840 / 480, 3528 / 441
            TH_DYN_CASE(11, 1)  // t_overlap 16: 3840 / 240, 3528 / 220
            TH_DYN_CASE(11, 0)  // t_overlap 32: 3840 / 120, 3528 / 110
            TH_DYN_CASE(11, 13) // 88.2 kHz, t_overlap 2: 3528 / 1764 (96 kHz, 3840 / 1920, k = 15: measured no gain at 7 waves, not instantiated)
#undef TH_DYN_CASE
        }
        return hipErrorInvalidValue;
    }
    // (the moment-form epilogue of n_fft 1024 / 2048 is instantiated for the full-reload frame loop only)
    const int sh = (LOG2_NC <= 10 && out.mode == 2 && out.mel_moment != 0) ? 0 : wave_shift<LOG2_NC>(g);
    // instantiate the common overlaps only: 75 % (hop = n_fft/4), 50 % and 87.5 %
#define TH_SHIFT_CASE(SH)                                                                                             \
    if constexpr ((SH) > 0 && (SH) < P)                                                                               \
        if (sh == (SH))                                                                                               \
            return launch_wave_t4<LOG2_NC, WAVES, (SH)>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, \
                                                                d_minmax, d_queue_head, n_cu, out, s);
    TH_SHIFT_CASE(P / 4)
    TH_SHIFT_CASE(P / 2)
    TH_SHIFT_CASE(P / 8)
#undef TH_SHIFT_CASE
    return launch_wave_t4<LOG2_NC, WAVES, 0>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax,
                                                     d_queue_head, n_cu, out, s);
}

template <int LOG2_NC>
static hipError_t launch_wave_t2(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                                 uint32_t n_chan, uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw,
                                 float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, int waves, const WaveOut &out, hipStream_t s) {
#define TH_WAVE_CASE(WV)                                                                                         \
    case WV:                                                                                                     \
        return launch_wave_t3<LOG2_NC, WV>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, \
                                                   d_queue_head, n_cu, out, s);
#if TH_AB_VARIANTS
    switch (waves) {
        TH_WAVE_CASE(4)
        TH_WAVE_CASE(6)
        TH_WAVE_CASE(7)
        TH_WAVE_CASE(8)
        TH_WAVE_CASE(10)
        TH_WAVE_CASE(12)
        TH_WAVE_CASE(14)
        TH_WAVE_CASE(16)
        default: return hipErrorInvalidValue;
    }
#else
    // the product build: each size's own launch shapes only (n_fft 1024 / 2048: 12 waves; n_fft 4096: 8, and 7 for the grid-aligned
    // mode with two window tables) — the other wave counts are tuning shapes of A/B builds (kernels.h)
    switch (waves) {
        case WaveLaunchCfg<LOG2_NC>::DEFAULT_WAVES:
            return launch_wave_t3<LOG2_NC, WaveLaunchCfg<LOG2_NC>::DEFAULT_WAVES>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
        default:
            if constexpr (WaveLaunchCfg<LOG2_NC>::GRID_WAVES != WaveLaunchCfg<LOG2_NC>::DEFAULT_WAVES)
                if (waves == WaveLaunchCfg<LOG2_NC>::GRID_WAVES)
                    return launch_wave_t3<LOG2_NC, WaveLaunchCfg<LOG2_NC>::GRID_WAVES>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, out, s);
            return hipErrorInvalidValue;
    }
#endif
#undef TH_WAVE_CASE
}

template <int LOG2_NC>
static hipError_t launch_wave_t(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                                uint32_t n_chan, uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw,
                                float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, int waves, const WaveOut &out, hipStream_t s) {
    if (waves <= 0) waves = WaveLaunchCfg<LOG2_NC>::DEFAULT_WAVES;
    return launch_wave_t2<LOG2_NC>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu,
                                   waves, out, s);
}

// this size's launcher (the dispatcher in the main translation unit calls it)
#define TH_PART_NAME2(N) launch_stft_wave_part_##N
#define TH_PART_NAME(N) TH_PART_NAME2(N)
hipError_t TH_PART_NAME(TH_STFT_PART)(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan, uint32_t n_tiles,
                                      const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, int waves,
                                      const WaveOut &out, hipStream_t s) {
    return launch_wave_t<TH_STFT_PART>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s);
}
#endif  // !TH_PART_MAIN
#if TH_PART_MAIN
hipError_t launch_stft_wave_part_9(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan, uint32_t n_tiles,
                                   const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, int waves,
                                   const WaveOut &out, hipStream_t s);
hipError_t launch_stft_wave_part_10(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan, uint32_t n_tiles,
                                    const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, int waves,
                                    const WaveOut &out, hipStream_t s);
hipError_t launch_stft_wave_part_11(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan, uint32_t n_tiles,
                                    const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax, uint32_t *d_queue_head, uint32_t n_cu, int waves,
                                    const WaveOut &out, hipStream_t s);
// grid-aligned register reuse for hops that are not multiples of 128 samples (stft_wave_kernel): 0 = not applicable, 1 =
// phased (n_fft 2048, hop = 3 * 128 + 96, n_fft - win >= 96), 2 = dynamic (n_fft 2048: any other hop in (256, 512), n_fft
// 1024: hop in (128, 256); n_fft - win >= 127); default launch shapes, dB output
int stft_wave_phased_mode(const StftGeom &g, int waves) {
    if ((g.win & 1u) || g.hop % 128 == 0) return 0;
    const uint32_t k = g.hop / 128;
    if (g.log2_nc == 10 && (waves <= 0 || waves == WaveLaunchCfg<10>::DEFAULT_WAVES)) {
        if (g.hop == 3 * 128 + 96 && g.n_fft - g.win >= 96) return 1;     // phased: rotation
        // dynamic: moves.  k = 2, 3: the 44.1 / 32 kHz defaults; k = 1, 0: t_overlap 8, 16, 32 at the 40 ms default (hop 240,
        // 120, 60 — tracks.ts:207), where a frame brings in two, one or no new 128-sample slot
        // k = 6, 7: t_overlap 2 at the 40 ms default (44.1 kHz: 1764 / 882, 48 kHz: 1920 / 960): half of the slots are new
        if ((k <= 3 || k == 6 || k == 7) && g.n_fft - g.win >= 127) return 2;
    }
    if (g.log2_nc == 9 && (waves <= 0 || waves == WaveLaunchCfg<9>::DEFAULT_WAVES) && k <= 1 && g.n_fft - g.win >= 127)
        return 2;  // (k = 0: 16 kHz with t_overlap 8 .. 32, 640 / 80 / 1024)
    // n_fft 4096 (the 40 ms default at 88.2 / 96 kHz): 7 waves per workgroup, the eighth's LDS holds the second window table
    // (3: hop and win / 2 both even — the frame's offset above the grid is then always even, the odd window table is never
    // read and its 17 KB hold the eighth wave: the 96 kHz (3840 / 960) and 88.2 kHz (3528 / 882) defaults and every even hop below)
    if (TH_DYN_EVEN_4096 && g.log2_nc == 11 && (waves <= 0 || waves == WaveLaunchCfg<11>::DEFAULT_WAVES) && (k == 7 || k == 6 || k == 3 || k <= 1 || k == 13) && g.hop % 2 == 0 &&
        (g.win / 2) % 2 == 0 && g.n_fft - g.win >= 127)
        return 3;
    if (g.log2_nc == 11 && (waves <= 0 || waves == WaveLaunchCfg<11>::GRID_WAVES) && (k == 7 || k == 6 || k == 3 || k <= 1 || k == 13) && g.n_fft - g.win >= 127)
        return 2;
    return 0;
}

uint32_t stft_wave_mel_max_pieces(const StftGeom &g) {
    if (g.log2_nc != 9 && g.log2_nc != 10) return 0;  // n_fft = 1024, 2048 (4096 is LDS-bound: no room for the table)
    if (g.log2_nc == 9) return MEL_PRF_1024;                              // its own LDS region per wave
    const uint32_t slab_cf32 = g.nc + g.nc / 16, prf0 = (g.nc + 2) / 2;  // WaveFft::SLAB_LEN, first (r, f) slot
    const uint32_t cap = (slab_cf32 - prf0) / 64 * 64;                    // whole slots of 64 pieces
    return cap;
}
bool stft_wave_mel_fits(const StftGeom &g, int waves, uint32_t words, bool banded) {
    const size_t extra = (size_t)words * 4;
    if (g.log2_nc == 10)
        return (waves <= 0 || waves == WaveLaunchCfg<10>::DEFAULT_WAVES) &&
               wave_lds_bytes<10, WaveLaunchCfg<10>::DEFAULT_WAVES>() + extra + 48 * sizeof(cf32) <= 160 * 1024;  // (+ phased pad)
    if (g.log2_nc == 9)
        return (waves <= 0 || waves == WaveLaunchCfg<9>::DEFAULT_WAVES) &&
               wave_lds_bytes<9, WaveLaunchCfg<9>::DEFAULT_WAVES>() + extra + (512 + 128) * sizeof(cf32) +
                       (banded ? 0 : (size_t)WaveLaunchCfg<9>::DEFAULT_WAVES * MEL_PRF_1024 * sizeof(cf32)) <= 160 * 1024;
    // n_fft 4096: nothing to fit (the moment form keeps no table in LDS); the launch shapes the epilogue is instantiated for
    // (every hop: the plain frame loop takes any — reuse of 8 / 16 / 4 slots or none; of the grid-aligned loops the epilogue is instantiated for
    // those stft_wave_mel_grid_ok names, a mel plan elsewhere runs the plain loop: stft_wave_mel_phase_mode)
    if (g.log2_nc == 11 && banded && words > 0 && (waves <= 0 || waves == WaveLaunchCfg<11>::DEFAULT_WAVES)) return true;
    return false;
}
// n_fft 4096 with the mel epilogue: the frame loop's mode.  The grid-aligned loop (stft_wave_phased_mode) where the epilogue is
// instantiated for it — the even-offset shapes of the 96 / 88.2 kHz defaults at t_overlap 4, 2 and 8 — else the plain loop: a
// mel plan at hop 240 / 120 / 441 ... gives up the grid's register reuse (7 % of the linear kernel) and keeps one kernel
// instead of two (round 6: 48 kHz 1920 / 240 / 4096, 695 mels).
int stft_wave_mel_phase_mode(const StftGeom &g, int waves) {
    const int pm = stft_wave_phased_mode(g, waves);
    if (g.log2_nc != 11) return pm;
    return pm == 3 ? 3 : 0;  // (every even-offset shape stft_wave_phased_mode names: hop / 128 = 7, 6, 3, 1, 0, 13)
}
bool stft_wave_mel_pair_applies(const StftGeom &g, int waves, uint32_t reach) {
    // (mel_banded_pair and wave_mel_flush read the table's PAIRED layout: a -DTH_MEL_BAND_PAIRED=0 build keeps the one-frame epilogue — ADVICE r5)
    if (TH_MEL_BAND_PAIRED == 0) return false;
    // (any frame loop of n_fft 1024 / 2048: plain, rotating, phased, dynamic; the default launch shape)
    if (g.log2_nc == 10) return (waves <= 0 || waves == WaveLaunchCfg<10>::DEFAULT_WAVES) && reach <= g.n_freq + (uint32_t)MelPair<10>::PAD;
    if (g.log2_nc == 9) return (waves <= 0 || waves == WaveLaunchCfg<9>::DEFAULT_WAVES) && reach <= g.n_freq + (uint32_t)MelPair<9>::PAD;
    return false;  // reads must stay inside the zeros behind each of the two amplitude rows
}
// (the same sum as launch_wave_multi_n; ADVICE r3: without this check a small growth of SLAB_LEN or MEL_ROWS_W would turn the
// default path into hipErrorInvalidValue at launch instead of the two-kernel fallback)
uint32_t stft_wave_multi_amp_pitch() { return (uint32_t)MELR_AP; }
bool stft_wave_multi_mel_fits(const StftGeom &g, int waves, uint32_t words) {
    if (g.log2_nc != 8) return false;
    using W = WaveFftM<8>;
    const int wv = waves > 0 ? waves : stft_wave_default_waves(g);
    return sizeof(cf32) * ((size_t)2 * W::NC + W::T2_LEN + W::T3_LEN + (size_t)wv * W::SLAB_LEN) + (size_t)words * 4 <= 160 * 1024;
}

bool stft_wave_sweep_applies(const StftGeom &g, int waves, int out_mode) {
    return (TH_AB_VARIANTS != 0) && g.log2_nc == 10 && g.phased == 0 && g.hop * 4 == g.n_fft && out_mode == 0 && (TH_USE_PK == 0) &&
           (waves <= 0 || waves == WaveLaunchCfg<10>::DEFAULT_WAVES);
}

int stft_wave_default_waves(const StftGeom &g) {
    switch (g.log2_nc) {
        case 11: return (g.phased && g.phased != 3) ? WaveLaunchCfg<11>::GRID_WAVES : WaveLaunchCfg<11>::DEFAULT_WAVES;
        case 10: return WaveLaunchCfg<10>::DEFAULT_WAVES;
        // n_fft 512 (four frames per wave, staged loads): measured 12 / 2 x 8 / 16 waves per CU — 512/128: 0.77 / 0.70-0.74 /
        // 0.68-0.72 ms, 320/80 (8 kHz default): 1.16-1.18 / 1.21-1.26 / 1.24 ms
        case 8: return g.hop >= 128 ? 16 : 12;
        case 12: case 13: case 14: return 12;  // (block kernel: only sizes the chunks, 12 x CUs of them per round)
        default: return WaveLaunchCfg<9>::DEFAULT_WAVES;
    }
}

hipError_t launch_stft_wave(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan,
                            uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax,
                            uint32_t *d_queue_head, uint32_t n_cu, int waves, const WaveOut &out, hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    if (!d_queue_head) return hipErrorInvalidValue;
    if (out.multi) {  // several short frames per wave (stft_wave_multi.h)
        if (!stft_wave_multi_applies(g, out.mode)) return hipErrorInvalidValue;
        if (out.mode == 1)  // amplitude rows (mel on the matrix cores): n_fft 512 only, 1024 has the one-frame kernel for that
            return g.log2_nc == 8 ? launch_wave_multi<8, 1>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, nullptr, d_queue_head, n_cu, waves, out, s)
                                  : hipErrorInvalidValue;
        if (out.mode == 2 && out.mel_moment != 0)  // mel rows in the moment form (filters wider than the banded table's 8 bins): its own instantiation
            return g.log2_nc == 8 ? launch_wave_multi<8, 4>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s)
                                  : hipErrorInvalidValue;
        if (out.mode == 2)  // mel rows by banded sums in the epilogue (n_fft 512 under narrow filters)
            return g.log2_nc == 8 ? launch_wave_multi<8, 2>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s)
                                  : hipErrorInvalidValue;
        if (g.log2_nc == 8) return launch_wave_multi<8, 0>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s);
        return launch_wave_multi<9, 0>(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s);
    }
    if (g.log2_nc >= 12) {  // one workgroup per frame (stft_block.h)
        if (g.phased) return hipErrorInvalidValue;
        if (out.mode == 2) {  // mel rows in the block kernel's epilogue, moment form (round 6): n_fft 8192 / 16384
            if (out.mel_tab == nullptr || out.mel_groups == 0 || !stft_block_mel_fused_applies(g, out.long_plan)) return hipErrorInvalidValue;
            if (g.log2_nc == 12) return launch_block<12, 2>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, d_minmax, s, &out);
            return launch_block<13, 2>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, d_minmax, s, &out);
        }
        if (out.mode > 1) return hipErrorInvalidValue;
        // round 5: R wave transforms + one combining pass (kernels_stft_long.hip): the default at n_fft 32768, selector 15 elsewhere
        if (stft_subwave_applies(g) && out.subwave_twc != nullptr && (out.long_plan == 2 || (out.long_plan == 0 && stft_subwave_default(g))))
            return launch_stft_subwave(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, out.subwave_twc, out.mode == 1 ? nullptr : d_minmax, out.mode == 1, n_cu, s);
        if (out.mode == 1) {  // amplitude rows, no (min, max)
            if (g.log2_nc == 12) return launch_block<12, 1>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, nullptr, s);
            if (g.log2_nc == 13) return launch_block<13, 1>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, nullptr, s);
#if TH_AB_VARIANTS  // (n_fft 32768 / 65536 run stft_subwave_kernel; their block kernels are selector 14 of A/B builds)
            if (g.log2_nc == 14) return launch_block<14, 1>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, nullptr, s);
            if (g.log2_nc == 15) return launch_block_planar<15, true>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, nullptr, s);
#endif
            return hipErrorInvalidValue;
        }
        if (g.log2_nc == 12) return launch_block<12, 0>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, d_minmax, s);
        if (g.log2_nc == 13) return launch_block<13, 0>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, d_minmax, s);
#if TH_AB_VARIANTS
        if (g.log2_nc == 14) return launch_block<14, 0>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, d_minmax, s);
        if (g.log2_nc == 15) return launch_block_planar<15, false>(g, d_jobs, d_tile_start, n_tiles, d_wtab, d_tw, d_minmax, s);
#endif
        return hipErrorInvalidValue;
    }
    switch (g.log2_nc) {
        case 9: return launch_stft_wave_part_9(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s);
        case 10: return launch_stft_wave_part_10(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s);
        case 11: return launch_stft_wave_part_11(g, d_jobs, d_tile_start, n_chan, n_tiles, d_wtab, d_tw, d_minmax, d_queue_head, n_cu, waves, out, s);
        default: return hipErrorInvalidValue;
    }
}

#endif  // TH_PART_MAIN
}  // namespace th
