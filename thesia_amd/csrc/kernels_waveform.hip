// kernels_waveform.hip — per-bin (min, max, mean) waveform decimator.
// Reference: encode_waveform_tile + waveform_bin_stats, src-tauri/src/core/render_tiles.rs:232-279.
// Bin b of a tile at level L covers samples [start + b*2^L, min(end, start + (b+1)*2^L)),
// end = min(N, start + 1024*2^L).  min/max use f32::min/max semantics (NaN-ignoring) and are
// exact under any order; the mean is sum/len in f32 — sequential for bins < 32 samples exactly
// as the reference's scalar branch (:270-278), lane-strided partial sums + shuffle tree for
// larger bins (the reference's own order there depends on its SIMD tier and pointer alignment,
// SURVEY.md A12).  Pure HBM streaming: 4 B read per sample.
#include <hip/hip_runtime.h>

#include <type_traits>

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

__device__ __forceinline__ uint64_t pyramid_dev_bins(uint64_t n, uint32_t level) {
    return level >= 63 ? (n ? 1 : 0) : (n + (1ull << level) - 1) >> level;
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


// ------------------------------------------------------------------------------------------
// Pyramid: every decimation level of a channel from ONE pass over the audio.
// Level L lives at float offset pyramid_offset(n, L) of the channel's output: ceil(n / 2^L) bins of
// (min, max, mean).  pyramid_base_kernel (one block per 4096 samples) reads the samples once and emits levels
// 0..4 (bins of <= 16 samples: the mean is the reference's sequential sum, render_tiles.rs:270-278, bit for
// bit), levels 5..12 by a pairwise tree over the block, and the level-12 sums; pyramid_up_kernel builds level
// L >= 13 from level L-1 (min of mins, max of maxes, sum of sums / len — the reference's own summation order for bins >= 32 samples depends on its SIMD tier and on
// pointer alignment; tolerance 1e-6 of the peak).
// HBM traffic per sample: 4 B read + 12 * (1 + 1/2 + ... ) = 24 B written.  The 12-byte bins of levels 0..4
// (97 % of the bytes) are staged through LDS so that every store instruction writes 256 B .. 1 KiB of
// contiguous bytes per wave.  (A first version built levels >= 5 with one launch per level from the level
// below: those eight small launches took as long as the base pass.)
// ------------------------------------------------------------------------------------------
uint64_t pyramid_bins(uint64_t n, uint32_t level) {
    if (n == 0) return 0;
    if (level >= 63) return 1;
    return (n + (1ull << level) - 1) >> level;
}
// Float offset of a level inside a channel's pyramid.  Every level starts on a 128-byte boundary (and the total is a
// multiple of 32 floats, so channels packed back to back stay aligned): level 0 is written with 16-byte stores laid on the
// output address, and a channel whose base was 4, 8 or 12 bytes off that grid took the dword-store fallback for half of
// all the bytes — with the dense layout three of four packed channels of the 13-level pyramid did (config 3: 2.10 ms;
// the same pass into an aligned 11-level pyramid 1.83 ms).
uint64_t pyramid_offset(uint64_t n, uint32_t level) {
    uint64_t off = 0;
    for (uint32_t l = 0; l < level; l++) off += (3 * pyramid_bins(n, l) + 31) / 32 * 32;
    return off;
}

constexpr uint32_t PYR_SPT = 16;                  // samples per thread
constexpr uint32_t PYR_SEG = 256 * PYR_SPT;       // samples per block
constexpr uint32_t PYR_LDS_STRIDE = 3 * PYR_SPT + 1;  // 49 dwords per lane: conflict-free staging

// stage `cnt` floats of this lane (v[0..cnt)) and write the wave's valid part as contiguous dwords
template <uint32_t CNT>
__device__ __forceinline__ void pyr_emit(float *stage, uint32_t lane, const float (&v)[3 * PYR_SPT], gptr<float> dst,
                                         uint64_t n_valid_dwords) {
    constexpr uint32_t S = (CNT % 2 == 0) ? CNT + 1 : CNT;  // odd stride: no LDS bank conflicts
#pragma unroll
    for (uint32_t j = 0; j < CNT; j++) stage[lane * S + j] = v[j];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (uint32_t i = 0; i < CNT; i++) {
        const uint32_t d = i * 64 + lane;          // dword of the wave's output for this level
        const uint32_t owner = d / CNT, j = d - owner * CNT;
        if (d < n_valid_dwords) dst[d] = stage[owner * S + j];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// grid: (segments of the longest channel, channels); shorter channels leave early
__global__ __launch_bounds__(256) void pyramid_base_kernel(const PyrJob *__restrict__ jobs) {
    __shared__ float stage_all[4][64 * PYR_LDS_STRIDE];
    __shared__ float wtot[4][3];
    __shared__ uint32_t waves_done;
    if (threadIdx.x == 0) waves_done = 0;
    __syncthreads();  // (at the start, where the four waves are still in step: costs nothing)
    const PyrJob job = jobs[blockIdx.y];
    // Workgroups are dealt round-robin over the 8 XCDs: give every XCD a contiguous run of segments instead of every
    // eighth one (the segment may then lie past the channel's end: checked right below).  Measured on config 3: levels 0
    // only 1.08 -> 0.98 ms (6.0 TB/s), all levels 2.15 -> 2.05 ms — each XCD then streams through its own region of the
    // output (and the small upper-level bins that share a 128-byte line are written through one L2, not eight).
    const uint32_t per_xcd = gridDim.x / 8;  // the launch rounds the grid up to a multiple of 8: the map is a bijection
    const uint32_t seg = (blockIdx.x % 8) * per_xcd + blockIdx.x / 8;
    if ((uint64_t)seg * PYR_SEG >= job.n_samples) return;
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    float *stage = stage_all[wv];
    const gptr<const float> wav = as_global(job.wav);
    const uint64_t n = job.n_samples;
    const uint64_t wave_base = (uint64_t)seg * PYR_SEG + (uint64_t)wv * 64 * PYR_SPT;  // first sample of this wave
    const uint64_t base = wave_base + (uint64_t)lane * PYR_SPT;
    const uint32_t valid = base >= n ? 0u : (uint32_t)min((uint64_t)PYR_SPT, n - base);
    float x[PYR_SPT];
    if (valid == PYR_SPT && (job.aligned16 & 1u)) {
#pragma unroll
        for (uint32_t q = 0; q < PYR_SPT / 4; q++) {
            const float4 t = *reinterpret_cast<gptr<const float4>>(wav + base + 4 * q);
            x[4 * q] = t.x;
            x[4 * q + 1] = t.y;
            x[4 * q + 2] = t.z;
            x[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < PYR_SPT; i++) x[i] = i < valid ? wav[base + i] : 0.0f;
    }
    const gptr<float> out = as_global(job.out);
    float v[3 * PYR_SPT];
    // level 0: one-sample bins (min = max = mean = x), half of all the bytes this kernel writes.  Output dword d
    // of the wave is sample d/3, so a lane builds whole 16-byte groups from two neighbouring samples read back
    // from LDS (samples staged with one pad dword per lane) and stores them as float4: 1 KiB per wave-instruction.
    const uint32_t first_level = job.aligned16 >> 1;  // levels below it are not materialised (th_pyramid_desc.first_level)
    if (job.n_levels > 0 && first_level == 0) {
        const uint64_t first = wave_base;  // bin index of the wave's first bin at level 0
        const uint64_t nb = pyramid_dev_bins(n, 0);
        const uint64_t vd = first >= nb ? 0 : 3 * min((uint64_t)64 * PYR_SPT, nb - first);  // valid dwords
#pragma unroll
        for (uint32_t i = 0; i < PYR_SPT; i++) stage[lane * (PYR_SPT + 1) + i] = x[i];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const gptr<float> dst = out + 3 * first;
        const bool al = ((reinterpret_cast<uintptr_t>(job.out) + 12 * first) & 15u) == 0;  // wave-uniform
#pragma unroll
        for (uint32_t i = 0; i < 3 * PYR_SPT / 4; i++) {
            const uint32_t q = i * 64 + lane, d = 4 * q;  // float4 group q = dwords d .. d+3
            const uint32_t s0 = d / 3, r = d - 3 * s0;    // dword d is sample s0, component r
            const uint32_t s1 = min(s0 + 1, 64 * PYR_SPT - 1);
            const float a = stage[s0 + s0 / PYR_SPT], b = stage[s1 + s1 / PYR_SPT];
            // r = 0: a a a b   r = 1: a a b b   r = 2: a b b b
            const float4 o = make_float4(a, r == 2 ? b : a, r == 0 ? a : b, b);
            if (d + 4 <= vd && al) {
                *reinterpret_cast<gptr<float4>>(dst + d) = o;
            } else {
                if (d < vd) dst[d] = o.x;
                if (d + 1 < vd) dst[d + 1] = o.y;
                if (d + 2 < vd) dst[d + 2] = o.z;
                if (d + 3 < vd) dst[d + 3] = o.w;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    float sum16 = 0.0f, mn16 = __builtin_inff(), mx16 = -__builtin_inff();
    // levels 1..4: bins of 2, 4, 8, 16 samples, sequential sums exactly as render_tiles.rs:270-278
#define TH_PYR_LEVEL(L)                                                                                        \
    {                                                                                                          \
        constexpr uint32_t SPB = 1u << (L), NB = PYR_SPT >> (L);                                               \
        _Pragma("unroll") for (uint32_t b = 0; b < NB; b++) {                                                  \
            float mn = __builtin_inff(), mx = -__builtin_inff(), sum = 0.0f;                                   \
            uint32_t len = 0;                                                                                  \
            _Pragma("unroll") for (uint32_t i = 0; i < SPB; i++) {                                             \
                const uint32_t k = b * SPB + i;                                                                \
                if (k < valid) {                                                                               \
                    mn = fminf(mn, x[k]);                                                                      \
                    mx = fmaxf(mx, x[k]);                                                                      \
                    sum = sum + x[k];                                                                          \
                    len++;                                                                                     \
                }                                                                                              \
            }                                                                                                  \
            v[3 * b] = mn;                                                                                     \
            v[3 * b + 1] = mx;                                                                                 \
            v[3 * b + 2] = sum / (float)len;                                                                   \
            if ((L) == 4) {                                                                                    \
                sum16 = sum;                                                                                   \
                mn16 = mn;                                                                                     \
                mx16 = mx;                                                                                     \
            }                                                                                                  \
        }                                                                                                      \
        if (job.n_levels > (L) && first_level <= (L)) {                                                        \
            const uint64_t first = wave_base >> (L);                                                           \
            const uint64_t nb = pyramid_dev_bins(n, (L));                                                      \
            const uint64_t vd = first >= nb ? 0 : 3 * min((uint64_t)64 * NB, nb - first);                      \
            pyr_emit<3 * NB>(stage, lane, v, out + job.level_off[(L)] + 3 * first, vd);                        \
        }                                                                                                      \
    }
    TH_PYR_LEVEL(1)
    TH_PYR_LEVEL(2)
    TH_PYR_LEVEL(3)
    TH_PYR_LEVEL(4)
#undef TH_PYR_LEVEL
    // levels 5..10 (bins of 32 .. 1024 samples = 2 .. 64 lanes): xor-butterfly inside the wave — min of mins, max
    // of maxes, pairwise sums; the first lane of every group writes the bin.  (12-byte strided stores: these
    // levels together are 3 % of the bytes.)
    // (only the first lane of a group of 2^k lanes keeps a meaningful value, and it needs the lane 2^(k-1) above it:
    // inside a 16-lane row that is a DPP row shift at register speed — `row_shl:n` gives lane i the value of lane i + n —
    // and only the two cross-row steps go through ds_bpermute; commutative operations, so the result is bit for bit the
    // xor-butterfly's)
    float mn = mn16, mx = mx16, sum = sum16;
    auto up = [](float v, auto kc) -> float {
        constexpr uint32_t K = decltype(kc)::value;
        if constexpr (K <= 4) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x100 + (1 << (K - 1)), 0xf, 0xf, false));
        } else {
            return __shfl_down(v, 1 << (K - 1), 64);
        }
    };
#pragma unroll
    for (uint32_t kk = 1; kk <= 6; kk++) {
        const uint32_t k = kk;
        const uint32_t level = 4 + k;
        float pmn, pmx, psum;
        switch (kk) {  // (the DPP control word must be a compile-time constant)
            case 1: pmn = up(mn, std::integral_constant<uint32_t, 1>{}); pmx = up(mx, std::integral_constant<uint32_t, 1>{}); psum = up(sum, std::integral_constant<uint32_t, 1>{}); break;
            case 2: pmn = up(mn, std::integral_constant<uint32_t, 2>{}); pmx = up(mx, std::integral_constant<uint32_t, 2>{}); psum = up(sum, std::integral_constant<uint32_t, 2>{}); break;
            case 3: pmn = up(mn, std::integral_constant<uint32_t, 3>{}); pmx = up(mx, std::integral_constant<uint32_t, 3>{}); psum = up(sum, std::integral_constant<uint32_t, 3>{}); break;
            case 4: pmn = up(mn, std::integral_constant<uint32_t, 4>{}); pmx = up(mx, std::integral_constant<uint32_t, 4>{}); psum = up(sum, std::integral_constant<uint32_t, 4>{}); break;
            case 5: pmn = up(mn, std::integral_constant<uint32_t, 5>{}); pmx = up(mx, std::integral_constant<uint32_t, 5>{}); psum = up(sum, std::integral_constant<uint32_t, 5>{}); break;
            default: pmn = up(mn, std::integral_constant<uint32_t, 6>{}); pmx = up(mx, std::integral_constant<uint32_t, 6>{}); psum = up(sum, std::integral_constant<uint32_t, 6>{}); break;
        }
        mn = fminf(mn, pmn);
        mx = fmaxf(mx, pmx);
        sum = sum + psum;
        const uint64_t s0 = wave_base + (uint64_t)(lane >> k << k) * PYR_SPT;  // first sample of the group's bin
        if (level < job.n_levels && (lane & ((1u << k) - 1)) == 0 && s0 < n) {
            const uint64_t len = min(n, s0 + ((uint64_t)PYR_SPT << k)) - s0;
            const gptr<float> d = out + job.level_off[level] + 3 * (s0 >> level);
            d[0] = mn;
            d[1] = mx;
            d[2] = sum / (float)len;
        }
    }
    // levels 11, 12 (2 and 4 waves) through LDS; the level-12 sums feed pyramid_up_kernel for levels >= 13.  No
    // workgroup barrier here: every wave leaves its totals and counts itself (LDS atomic), and the wave that arrives last
    // writes the three bins — the other three are gone by then instead of waiting (the barrier version spent 7 % of the
    // kernel on these 36 bytes per block).
    if (lane == 0) {
        wtot[wv][0] = mn;
        wtot[wv][1] = mx;
        wtot[wv][2] = sum;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    uint32_t arrived = 0;
    if (lane == 0) arrived = atomicAdd(&waves_done, 1u);
    arrived = __builtin_amdgcn_readfirstlane(arrived);
    if (arrived != 3) return;
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    if (lane < 2) {  // level 11: waves (0,1) and (2,3)
        const uint32_t a = 2 * lane;
        const uint64_t s0 = (uint64_t)seg * PYR_SEG + (uint64_t)a * 64 * PYR_SPT;
        if (11 < job.n_levels && s0 < n) {
            const uint64_t len = min(n, s0 + 2048) - s0;
            const gptr<float> d = out + job.level_off[11] + 3 * (s0 >> 11);
            d[0] = fminf(wtot[a][0], wtot[a + 1][0]);
            d[1] = fmaxf(wtot[a][1], wtot[a + 1][1]);
            d[2] = (wtot[a][2] + wtot[a + 1][2]) / (float)len;
        }
    } else if (lane == 2) {  // level 12: the whole block
        const uint64_t s0 = (uint64_t)seg * PYR_SEG;
        const float s = (wtot[0][2] + wtot[1][2]) + (wtot[2][2] + wtot[3][2]);
        if (12 < job.n_levels) {
            const uint64_t len = min(n, s0 + 4096) - s0;
            const gptr<float> d = out + job.level_off[12] + 3 * (uint64_t)seg;
            d[0] = fminf(fminf(wtot[0][0], wtot[1][0]), fminf(wtot[2][0], wtot[3][0]));
            d[1] = fmaxf(fmaxf(wtot[0][1], wtot[1][1]), fmaxf(wtot[2][1], wtot[3][1]));
            d[2] = s / (float)len;
        }
        if (job.sums) as_global(job.sums)[seg] = s;
    }
}

// level L (>= 13) from level L-1: thread per destination bin (a few bins per channel)
__global__ __launch_bounds__(256) void pyramid_up_kernel(const PyrJob *__restrict__ jobs, uint32_t level,
                                                         uint32_t parity) {
    // (a reference, not a copy: level_off is indexed with the run-time level, and a by-value struct indexed dynamically
    // is put into scratch memory)
    const PyrJob &job = jobs[blockIdx.y];
    const uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t n = job.n_samples;
    const uint64_t nb = pyramid_dev_bins(n, level);
    if (b >= nb || level >= job.n_levels) return;
    const uint64_t nsrc = pyramid_dev_bins(n, level - 1);
    const gptr<const float> src = as_global(job.out) + job.level_off[level - 1];
    // sums ping-pong between the two halves of the scratch: level-12 sums sit in half 0
    const gptr<float> sums = as_global(job.sums);
    const uint64_t half = job.sums_half;
    const gptr<const float> ssrc = sums + (parity ? half : 0);
    const gptr<float> sdst = sums + (parity ? 0 : half);
    const uint64_t a = 2 * b;
    float mn = src[3 * a], mx = src[3 * a + 1], sum = ssrc[a];
    if (a + 1 < nsrc) {
        mn = fminf(mn, src[3 * (a + 1)]);
        mx = fmaxf(mx, src[3 * (a + 1) + 1]);
        sum = sum + ssrc[a + 1];
    }
    const uint64_t s0 = b << level;
    const uint64_t len = min(n, s0 + (1ull << level)) - s0;
    const gptr<float> dst = as_global(job.out) + job.level_off[level] + 3 * b;
    dst[0] = mn;
    dst[1] = mx;
    dst[2] = sum / (float)len;
    sdst[b] = sum;
}

hipError_t launch_pyramid_base(const PyrJob *d_jobs, uint32_t n_jobs, uint64_t max_samples, hipStream_t s) {
    const uint64_t segs = (max_samples + PYR_SEG - 1) / PYR_SEG;
    if (!n_jobs || !segs) return hipSuccess;
    const uint32_t grid_x = (uint32_t)((segs + 7) / 8 * 8);  // multiple of 8: see the segment order in the kernel
    hipLaunchKernelGGL(pyramid_base_kernel, dim3(grid_x, n_jobs), dim3(256), 0, s, d_jobs);
    return hipGetLastError();
}
hipError_t launch_pyramid_up(const PyrJob *d_jobs, uint32_t n_jobs, uint64_t max_samples, uint32_t level, uint32_t parity,
                             hipStream_t s) {
    const uint64_t blocks = (pyramid_bins(max_samples, level) + 255) / 256;
    if (!n_jobs || !blocks) return hipSuccess;
    hipLaunchKernelGGL(pyramid_up_kernel, dim3((uint32_t)blocks, n_jobs), dim3(256), 0, s, d_jobs, level, parity);
    return hipGetLastError();
}


// ------------------------------------------------------------------------------------------
// Channel statistics upstream of the path: sum of squares and absolute peak of a channel in one pass
// (reference: simd.rs:113-183 — sum_squares (Kahan-compensated f32, :820-832) and abs_max (:935-937), used by
// StatCalculator::calc, dynamics/stats.rs:56-86, for rms_dB and max_peak).  The sum is accumulated in f64
// (per-thread, wave shuffle tree, one f64 atomic per block) and rounded to f32 once: within 1 ulp of what the
// reference's compensated sums deliver, whatever their SIMD tier; the peak is exact (max of |x|).
// Pure HBM streaming: 4 B read per sample.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void channel_stats_kernel(const StatsJob *__restrict__ jobs, double *__restrict__ sumsq,
                                                            uint32_t *__restrict__ peak_bits) {
    __shared__ double wsum[4];
    __shared__ float wmax[4];
    const StatsJob job = jobs[blockIdx.y];
    const uint64_t base = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if ((uint64_t)blockIdx.x * 4096 >= job.n_samples) return;
    const gptr<const float> wav = as_global(job.wav);
    double acc = 0.0;
    float pk = 0.0f;
    if (base + 16 <= job.n_samples && job.aligned16) {
#pragma unroll
        for (uint32_t q = 0; q < 4; q++) {
            const float4 t = *reinterpret_cast<gptr<const float4>>(wav + base + 4 * q);
            acc += (double)t.x * (double)t.x + (double)t.y * (double)t.y + ((double)t.z * (double)t.z + (double)t.w * (double)t.w);
            pk = fmaxf(fmaxf(pk, fmaxf(fabsf(t.x), fabsf(t.y))), fmaxf(fabsf(t.z), fabsf(t.w)));
        }
    } else {
        for (uint32_t i = 0; i < 16; i++) {
            if (base + i < job.n_samples) {
                const float v = wav[base + i];
                acc += (double)v * (double)v;
                pk = fmaxf(pk, fabsf(v));  // f32::max ignores NaN (abs_max_scalar, :935-937)
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        pk = fmaxf(pk, __shfl_xor(pk, o, 64));
    }
    const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
    if (lane == 0) {
        wsum[wv] = acc;
        wmax[wv] = pk;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicAdd(&sumsq[blockIdx.y], (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
        // |x| >= 0: the IEEE bit patterns of non-negative floats order like unsigned integers
        atomicMax(&peak_bits[blockIdx.y], __float_as_uint(fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]))));
    }
}

hipError_t launch_channel_stats(const StatsJob *d_jobs, uint32_t n_jobs, uint64_t max_samples, double *d_sumsq,
                                uint32_t *d_peak_bits, hipStream_t s) {
    const uint64_t blocks = (max_samples + 4095) / 4096;
    if (!n_jobs || !blocks) return hipSuccess;
    hipLaunchKernelGGL(channel_stats_kernel, dim3((uint32_t)blocks, n_jobs), dim3(256), 0, s, d_jobs, d_sumsq, d_peak_bits);
    return hipGetLastError();
}

}  // namespace th
