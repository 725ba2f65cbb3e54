// kernels_stft.hip — fused STFT → |X| → [mel] → dB (+ per-channel min/max) kernels for gfx950.
//
// One launch processes a whole batch of independent channels (core/mod.rs:153-163).  A "tile" is
// `frames_per_tile` consecutive frames of one channel; blockIdx.x → tile through the per-channel
// prefix array tile_start[] (binary search, scalar loads).  Nothing but the f32 dB spec (which
// the reference retains, core/mod.rs:42) and 2 floats of min/max per channel is written to HBM:
// the windowed frames, the complex spectrum and the linear magnitudes never leave the CU.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"

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

__global__ void minmax_init_kernel(float *minmax, uint32_t n_chan) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_chan) {
        minmax[2 * i] = __builtin_inff();
        minmax[2 * i + 1] = -__builtin_inff();
    }
}

// f32::min / f32::max semantics of the reference's scalar reductions (NaN-ignoring).
__device__ __forceinline__ float nmin(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float nmax(float a, float b) { return fmaxf(a, b); }

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
// Generic workgroup kernel: any power-of-two n_fft in [8, 16384].  256 threads work on one frame
// at a time: LDS ping-pong Stockham radix-4 (+ one radix-2 pass when log2(Nc) is odd), split
// pass, optional banded mel reduction, dB, coalesced row store.  Correctness-first fallback for
// sizes the wave kernel does not cover.
// ------------------------------------------------------------------------------------------
constexpr int GEN_THREADS = 256;

__global__ __launch_bounds__(GEN_THREADS) void stft_generic_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ tile_start, uint32_t n_chan,
    const float *__restrict__ window, const cf32 *__restrict__ tw, const float *__restrict__ mel_fb,
    const uint32_t *__restrict__ mel_lo, const uint32_t *__restrict__ mel_hi, float *__restrict__ minmax) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf32 *bufA = reinterpret_cast<cf32 *>(smem_raw);
    cf32 *bufB = bufA + g.nc;
    __shared__ float red[2 * (GEN_THREADS / 64)];

    const uint32_t tid = threadIdx.x;
    const uint32_t chan = find_chan(tile_start, n_chan, blockIdx.x);
    const ChanJob job = jobs[chan];
    const uint32_t f0 = (blockIdx.x - tile_start[chan]) * g.frames_per_tile;
    const uint32_t f1 = min(f0 + g.frames_per_tile, job.n_frames);

    float lmin = __builtin_inff(), lmax = -__builtin_inff();

    for (uint32_t f = f0; f < f1; f++) {
        const int64_t s0 = (int64_t)f * g.hop - (int64_t)(g.win / 2);
        gen_load(tid, GEN_THREADS, g, job.wav, job.n_samples, s0, window, bufA);
        __syncthreads();
        cf32 *in = bufA, *out = bufB;
        uint32_t Ns = 1;
        if (g.log2_nc & 1) {
            gen_pass_r2(tid, GEN_THREADS, g, Ns, tw, in, out);
            __syncthreads();
            Ns = 2;
            cf32 *t = in; in = out; out = t;
        }
        for (; Ns < g.nc; Ns <<= 2) {
            gen_pass_r4(tid, GEN_THREADS, g, Ns, tw, in, out);
            __syncthreads();
            cf32 *t = in; in = out; out = t;
        }
        // `in` holds Z in natural order; magnitudes go to the other buffer (nc+1 floats fit in nc cf32)
        float *mag = reinterpret_cast<float *>(out);
        gen_split(tid, GEN_THREADS, g, tw, in, mag);
        __syncthreads();
        float *row = job.spec + (size_t)f * g.height;
        if (g.n_mel == 0) {
            for (uint32_t k = tid; k < g.n_freq; k += GEN_THREADS) {
                const float d = amp_to_dB(mag[k]);
                row[k] = d;
                lmin = nmin(lmin, d);
                lmax = nmax(lmax, d);
            }
        } else {
            // linspec.dot(mel_fb) restricted to each filter's non-zero band, ascending f
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
            atomic_min_f32(&minmax[2 * chan], a);
            atomic_max_f32(&minmax[2 * chan + 1], b);
        }
    }
}

// ------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------
hipError_t launch_minmax_init(float *d_minmax, uint32_t n_chan, hipStream_t s) {
    if (!d_minmax || !n_chan) return hipSuccess;
    hipLaunchKernelGGL(minmax_init_kernel, dim3((n_chan + 255) / 256), dim3(256), 0, s, d_minmax, n_chan);
    return hipGetLastError();
}

size_t stft_generic_lds_bytes(const StftGeom &g) { return 2 * (size_t)g.nc * sizeof(cf32); }

hipError_t launch_stft_generic(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                               uint32_t n_chan, uint32_t n_tiles, const float *d_window, const cf32 *d_tw,
                               const float *d_mel_fb, const uint32_t *d_mel_lo, const uint32_t *d_mel_hi,
                               float *d_minmax, hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    const size_t lds = stft_generic_lds_bytes(g);
    if (lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(stft_generic_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(stft_generic_kernel, dim3(n_tiles), dim3(GEN_THREADS), lds, s, g, d_jobs, d_tile_start,
                       n_chan, d_window, d_tw, d_mel_fb, d_mel_lo, d_mel_hi, d_minmax);
    return hipGetLastError();
}

}  // namespace th
