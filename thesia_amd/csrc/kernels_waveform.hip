// kernels_waveform.hip — per-bin (min, max, mean) waveform decimator.
// Reference: encode_waveform_tile + waveform_bin_stats, src-tauri/src/core/render_tiles.rs:232-279.
// Bin b of a tile at level L covers samples [start + b*2^L, min(end, start + (b+1)*2^L)),
// end = min(N, start + 1024*2^L).  min/max use f32::min/max semantics (NaN-ignoring) and are
// exact under any order; the mean is sum/len in f32 — sequential for bins < 32 samples exactly
// as the reference's scalar branch (:270-278), lane-strided partial sums + shuffle tree for
// larger bins (the reference's own order there depends on its SIMD tier and pointer alignment,
// SURVEY.md A12).  Pure HBM streaming: 4 B read per sample.
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"

// Integer / rounding stages must round once per operation exactly like the reference's scalar f32
// code: no FMA contraction anywhere in this file (the build also passes -ffp-contract=off).
#pragma clang fp contract(off)

namespace th {

__device__ __forceinline__ uint32_t find_wjob(const uint32_t *__restrict__ start, uint32_t n, uint32_t b) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= b) lo = mid;
        else hi = mid;
    }
    return lo;
}

constexpr uint32_t WAVE_SMALL_MAX_LEVEL = 5;  // spb <= 32: one thread per bin (sequential, like :270-278)

uint32_t waveform_blocks_for(uint32_t level, uint32_t bin_count) {
    if (!bin_count) return 0;
    if (level <= WAVE_SMALL_MAX_LEVEL) return (bin_count + 255) / 256;
    return (bin_count + 3) / 4;  // one wave per bin, 4 waves per block
}

__global__ __launch_bounds__(256) void waveform_kernel(const WaveJob *__restrict__ jobs,
                                                       const uint32_t *__restrict__ block_start, uint32_t n_jobs) {
    const uint32_t ji = find_wjob(block_start, n_jobs, blockIdx.x);
    const WaveJob job = jobs[ji];
    const gptr<const float> wav = as_global(job.wav);
    const uint32_t lb = blockIdx.x - block_start[ji];
    const uint64_t spb = job.level < 64 ? (1ull << job.level) : ~0ull;
    uint64_t tile_end = job.start + 1024ull * spb;  // callers keep level small enough not to overflow
    if (tile_end > job.n_samples || tile_end < job.start) tile_end = job.n_samples;

    if (job.level <= WAVE_SMALL_MAX_LEVEL) {
        const uint32_t b = lb * 256 + threadIdx.x;
        if (b >= job.bin_count) return;
        const uint64_t s = job.start + (uint64_t)b * spb;
        const uint64_t e = min(tile_end, s + spb);
        float mn = __builtin_inff(), mx = -__builtin_inff(), sum = 0.0f;
        for (uint64_t i = s; i < e; i++) {
            const float v = wav[i];
            mn = fminf(mn, v);
            mx = fmaxf(mx, v);
            sum = sum + v;
        }
        const gptr<float> o = as_global(job.bins) + 3ull * b;
        o[0] = mn;
        o[1] = mx;
        o[2] = sum / (float)(e - s);
        return;
    }

    const uint32_t lane = threadIdx.x & 63;
    const uint32_t b = lb * 4 + (threadIdx.x >> 6);
    if (b >= job.bin_count) return;
    const uint64_t s = job.start + (uint64_t)b * spb;
    const uint64_t e = min(tile_end, s + spb);
    float mn = __builtin_inff(), mx = -__builtin_inff(), sum = 0.0f;
    for (uint64_t i = s + lane; i < e; i += 64) {
        const float v = wav[i];
        mn = fminf(mn, v);
        mx = fmaxf(mx, v);
        sum += v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, o, 64));
        mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        sum += __shfl_xor(sum, o, 64);
    }
    if (lane == 0) {
        const gptr<float> o = as_global(job.bins) + 3ull * b;
        o[0] = mn;
        o[1] = mx;
        o[2] = sum / (float)(e - s);
    }
}

hipError_t launch_waveform(const WaveJob *d_jobs, const uint32_t *d_block_start, uint32_t n_jobs,
                           uint32_t n_blocks, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    hipLaunchKernelGGL(waveform_kernel, dim3(n_blocks), dim3(256), 0, s, d_jobs, d_block_start, n_jobs);
    return hipGetLastError();
}

}  // namespace th
