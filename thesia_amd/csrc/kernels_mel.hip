// kernels_mel.hip — mel filterbank contraction on the CDNA4 matrix cores.
//
// Reference: `linspec.dot(&mel_fb)` (src-tauri/src/core/spectrogram.rs:207; ndarray → OpenBLAS
// sgemm), then dB_from_amp_inplace_default (:208, dynamics/decibel.rs:170-214):
//     mel[t, m] = sum_f |X|[t, f] * fb[f, m]      (T x F) . (F x n_mel),  f32
// This is the path's one genuine dense contraction.  It runs on `v_mfma_f32_16x16x4_f32` (f32 in,
// f32 accumulate: bit-for-bit an fmaf chain, no reduced-precision shortcut — gfx950 has no xf32).
//
// Shape per workgroup: 4 waves x 16 frames = 64 frames; a wave computes all mel tiles (16 mels each) of its 16
// frames, one tile at a time (4 accumulator VGPRs).  K (= frequency bins) is walked in blocks of 16:
//   A operand: every lane loads ONE float4 of its frame row per K block (16 rows x 64 B per
//              wave-instruction); MFMA step s of the block uses element s, i.e. the block's 16 bins
//              are consumed in the order {s, 4+s, 8+s, 12+s} — a permutation of k, which a sum
//              does not care about — so no register shuffling is needed to feed the 16x16x4 shape.
//   B operand: the filterbank repacked per (tile, K block) in exactly the operand order, one 16-byte load per lane
//              per K block (1 KB contiguous per wave), L2-resident (80 KB at 128 mels, ~250 KB at 370).
//   Band structure: a mel filter is a triangle, so tile j only needs the K blocks [klo_j, khi_j)
//              (≈ 1/7 of the dense product at 128 mels).
// Epilogue: 20*log10 via v_log_f32, store, min/max, one atomic pair per workgroup.
// Input is the linear amplitude written by stft_wave_kernel<..., AMP = true> (columns >= n_freq of
// the amplitude buffer are zero-filled once, so the padded K tail multiplies 0 * 0).
// HBM traffic per frame: 4*n_freq B read + 4*n_mel B written (the amplitude round trip through HBM
// is the price of the two-kernel paths: n_fft 1024 / 2048 fuse the filterbank into the FFT kernel).
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "stft_core.h"
#include "stft_wave.h"  // mel_banded

namespace th {

using vf32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ uint32_t mel_find_job(const uint32_t *__restrict__ start, uint32_t n, uint32_t b) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= b) lo = mid;
        else hi = mid;
    }
    return lo;
}

__device__ __forceinline__ void mel_atomic_min(float *addr, float v) {
    if (__builtin_isnan(v)) return;
    if (!__builtin_signbit(v)) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void mel_atomic_max(float *addr, float v) {
    if (__builtin_isnan(v)) return;
    if (!__builtin_signbit(v)) atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// One wave = MEL_MT x 16 frames x all mel tiles (the MEL_MT row tiles share every B load).  Tile-outer / band-inner: for N tile j only the K blocks [klo_j, khi_j) hold
// non-zeros, so the inner loop is branch-free and is unrolled MEL_UNROLL K blocks deep with every load of the group
// issued before its first MFMA (the first version walked K outside with a wave-uniform band test per tile and exposed
// one HBM + one L2 latency per K block: 0.29 ms for config 4 where the amplitude stream alone takes 0.14 ms).
// B comes from a packed table: block (j, kb) = 64 lanes x 4 steps, laid out so that a lane's four B values are ONE
// 16-byte load and a wave's load is 1 KB contiguous; one all-zero block at index `zero_block` pads ragged groups.
// The amplitude rows of a band's first K blocks were read a moment ago as the previous tile's last ones (adjacent
// triangles overlap), so the ~1.2x re-read is served by L1/L2.
constexpr int MEL_UNROLL = 4;                       // K blocks per group = one 256-byte span of every row
constexpr int MEL_LDS_PITCH = 16 * MEL_UNROLL + 4;  // floats per row of the transpose tile
__device__ __forceinline__ void mel_wave_sync() {   // LDS hand-over inside one wave (no workgroup barrier)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__global__ __launch_bounds__(256) void mel_mfma_kernel(const MelJob *__restrict__ jobs,
                                                       const uint32_t *__restrict__ tile_start, uint32_t n_jobs,
                                                       uint32_t amp_pitch, const float *__restrict__ bt,
                                                       const uint32_t *__restrict__ tile_band,
                                                       const uint32_t *__restrict__ slice_start,
                                                       uint32_t zero_block, uint32_t n_mel,
                                                       float *__restrict__ minmax) {
    __shared__ float red[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ji = mel_find_job(tile_start, n_jobs, blockIdx.x);
    const MelJob job = jobs[ji];
    const uint32_t frame0 = job.f_begin + (blockIdx.x - tile_start[ji]) * MEL_TILE_FRAMES + wave * (16 * MEL_MT);
    const uint32_t kq = lane >> 4, li = lane & 15u;
    // rows past the end of the range are clamped for the loads and masked at the store
    // A operand: the MFMA layout wants lane (kq, li) to hold row li — read straight from memory that is 16 different
    // rows x 16 B per quarter wave, 64 cache-line accesses per load instruction, and the L1 tag rate (not HBM) bounds
    // the kernel (measured: loads alone 0.24 ms for 0.7 GB).  So the rows are loaded COALESCED (16 lanes x 16 B = 256
    // contiguous bytes of one row, 4 rows per instruction) and transposed through a wave-private LDS tile:
    // [MEL_MT][16 rows][64 + 4 floats] (the 4-float pad makes both the b128 writes and the b128 reads conflict-free).
    __shared__ __attribute__((aligned(16))) float lds_a[4][MEL_MT][16][MEL_LDS_PITCH];
    const uint32_t lr = lane >> 4, lq = lane & 15u;  // coalesced load: row 4c + lr of the tile, 16-byte column lq
    gptr<const float> ldrow[MEL_MT][4];
#pragma unroll
    for (int t = 0; t < MEL_MT; t++)
#pragma unroll
        for (int c = 0; c < 4; c++)
            ldrow[t][c] = as_global(job.amp) + (size_t)min(frame0 + 16 * t + 4 * c + lr, job.f_end - 1) * amp_pitch;
    const gptr<const vf32x4> bl = reinterpret_cast<gptr<const vf32x4>>(as_global(bt)) + lane;
    const gptr<float> spec = as_global(job.spec);
    float lmin = __builtin_inff(), lmax = -__builtin_inff();

    if (frame0 < job.f_end) {  // wave-uniform: this wave has at least one frame
        // blockIdx.y = slice of the mel tiles (slices hold about equal numbers of K groups): with every workgroup
        // walking all tiles a workgroup lives ~100 us (its groups are one dependent load latency each) and the 1.26
        // rounds the grid needs cost two; short work units fill the tail
        const uint32_t j0 = slice_start[blockIdx.y], j1 = slice_start[blockIdx.y + 1];
        for (uint32_t j = j0; j < j1; j++) {
            const uint32_t klo = tile_band[3 * j], khi = tile_band[3 * j + 1], off = tile_band[3 * j + 2];  // scalar loads
            vf32x4 acc[MEL_MT];
#pragma unroll
            for (int t = 0; t < MEL_MT; t++) acc[t] = vf32x4{0.f, 0.f, 0.f, 0.f};
            for (uint32_t kb = klo; kb < khi; kb += MEL_UNROLL) {
                vf32x4 raw[MEL_MT][4], b[MEL_UNROLL];
                // 64 floats of every row starting at K block kb; a ragged last group runs into the next K blocks (or, at
                // the end of the row, is clamped): whatever it reads there meets the all-zero B block
                const uint32_t col = min(16 * kb + 4 * lq, amp_pitch - 4);
#pragma unroll
                for (int t = 0; t < MEL_MT; t++)
#pragma unroll
                    for (int c = 0; c < 4; c++) raw[t][c] = *reinterpret_cast<gptr<const vf32x4>>(ldrow[t][c] + col);
#pragma unroll
                for (int u = 0; u < MEL_UNROLL; u++) {
                    const uint32_t kbi = kb + u < khi ? off + (kb + u - klo) : zero_block;  // wave-uniform
                    b[u] = bl[(size_t)kbi * 64];
                }
                mel_wave_sync();  // the previous group's reads of the tile are done
#pragma unroll
                for (int t = 0; t < MEL_MT; t++)
#pragma unroll
                    for (int c = 0; c < 4; c++) *reinterpret_cast<vf32x4 *>(&lds_a[wave][t][4 * c + lr][4 * lq]) = raw[t][c];
                mel_wave_sync();
                vf32x4 a[MEL_MT][MEL_UNROLL];
#pragma unroll
                for (int t = 0; t < MEL_MT; t++)
#pragma unroll
                    for (int u = 0; u < MEL_UNROLL; u++) a[t][u] = *reinterpret_cast<const vf32x4 *>(&lds_a[wave][t][li][16 * u + 4 * kq]);
#pragma unroll
                for (int u = 0; u < MEL_UNROLL; u++) {
#pragma unroll
                    for (int t = 0; t < MEL_MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].x, b[u].x, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < MEL_MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].y, b[u].y, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < MEL_MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].z, b[u].z, acc[t], 0, 0, 0);
#pragma unroll
                    for (int t = 0; t < MEL_MT; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t][u].w, b[u].w, acc[t], 0, 0, 0);
                }
            }
            // C/D layout of the 16x16 shapes: col = lane & 15, row = 4 * (lane >> 4) + r
            const uint32_t m = 16 * j + li;
#pragma unroll
            for (int t = 0; t < MEL_MT; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const uint32_t f = frame0 + 16 * t + 4 * kq + r;
                    if (f < job.f_end && m < n_mel) {
                        // dB_from_amp (decibel.rs:179-202): 20*log10(x); x = +0 -> -inf
                        const float d = 6.02059991327962390f * __builtin_amdgcn_logf(acc[t][r]);
                        spec[(size_t)f * job.spec_pitch + m] = d;
                        lmin = fminf(lmin, d);
                        lmax = fmaxf(lmax, d);
                    }
                }
        }
    }
    if (minmax != nullptr) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lmin = fminf(lmin, __shfl_xor(lmin, o, 64));
            lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
        }
        if (lane == 0) {
            red[2 * wave] = lmin;
            red[2 * wave + 1] = lmax;
        }
        __syncthreads();
        if (tid == 0) {
            float a = red[0], b = red[1];
            for (int w = 1; w < 4; w++) {
                a = fminf(a, red[2 * w]);
                b = fmaxf(b, red[2 * w + 1]);
            }
            mel_atomic_min(&minmax[2 * job.mm_index], a);
            mel_atomic_max(&minmax[2 * job.mm_index + 1], b);
        }
    }
}

// ------------------------------------------------------------------------------------------
// Short rows (n_fft 512: 257 bins) with mel counts near the bin count (the app's defaults at 8-12 kHz: 245-257 mels):
// mel_rows_kernel.  A filter is then 1-6 bins wide, a 16 x 16 block of the filterbank holds ~40 non-zeros, and the matrix
// cores spend 240 dependent 32-cycle MFMAs per 16 frames on them (measured: mel_mfma_kernel 2.2 ms for the 2.3 M frames of
// the 8 kHz default = 2.1 TB/s; a whole-row MFMA variant 1.39 ms, its loads of B ordered behind the row prefetch in vmcnt).
// Here the product is what it is, a banded sum: lane = mel (64 g + lane, g < n_groups <= 8), its W <= 8 weights and its
// first bin live in registers for the life of the (persistent) wave, and per frame a mel costs W LDS reads + W FMAs.
// A wave stages 16 whole amplitude rows in LDS (17 coalesced 16-byte loads per lane, the next 16 rows requested before
// this unit's sums start: nothing else of the unit loads from memory, so the prefetch really runs ahead) and stores every
// output row as 256-byte spans.  Sum order: ascending bins, one FMA each, from 0 (the generic kernel's order).
// tab: [n_groups][1 + W][64] words: first bin, then W weights (float bits; zero past the filter's last bin).
// ------------------------------------------------------------------------------------------
constexpr int MEL_ROWS_COLS = 16 * MEL_ROWS_NKB;      // columns staged per row
constexpr int MEL_ROWS_C4 = MEL_ROWS_COLS / 4;        // 16-byte columns per row
constexpr int MEL_ROWS_AP = MEL_ROWS_COLS + 4;        // LDS row pitch (floats)
constexpr int MEL_ROWS_NLD = (16 * MEL_ROWS_C4 + 63) / 64;
static_assert(16 * MEL_ROWS_C4 == 64 * MEL_ROWS_NLD, "the 16-row tile is a whole number of wave loads");
// (a filter's W reads start below n_freq and must end inside the row: the host checks n_freq - 1 + W <= MEL_ROWS_AP)
template <int W>
__global__ __launch_bounds__(256) void mel_rows_kernel(const MelJob *__restrict__ jobs, const uint32_t *__restrict__ tile_start,
                                                       uint32_t n_jobs, uint32_t n_tiles, uint32_t amp_pitch,
                                                       const uint32_t *__restrict__ tab, uint32_t n_groups, uint32_t n_mel,
                                                       float *__restrict__ minmax) {
    constexpr int MAXG = MEL_ROWS_MAX_GROUPS;
    __shared__ __attribute__((aligned(16))) float lds_a[4][16][MEL_ROWS_AP];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *const tile = &lds_a[wave][0][0];
    // Persistent waves: a "unit" is one 16-frame tile of a workgroup tile of the host's table (MEL_TILE_FRAMES frames =
    // 4 MEL_MT units); every wave takes a contiguous range of units, so it stays within a channel for long stretches
    // (one (min, max) atomic pair per channel it touches) and always has the next unit's rows in flight.
    constexpr uint32_t UPT = 4 * MEL_MT;
    const uint32_t n_units = n_tiles * UPT, n_w = gridDim.x * 4u, per = (n_units + n_w - 1) / n_w;
    const uint32_t q_end = min(n_units, (blockIdx.x * 4u + wave + 1u) * per);
    uint32_t q = (blockIdx.x * 4u + wave) * per;
    if (q >= q_end) return;
    uint32_t ji = mel_find_job(tile_start, n_jobs, q / UPT);
    // the next unit with frames at or after q: (job index, first frame); false past the wave's range
    auto locate = [&](uint32_t &qq, uint32_t &jj, uint32_t &frame0) {
        for (; qq < q_end; qq++) {
            const uint32_t t = qq / UPT;
            while (t >= tile_start[jj + 1]) jj++;
            frame0 = jobs[jj].f_begin + (t - tile_start[jj]) * MEL_TILE_FRAMES + 16u * (qq % UPT);
            if (frame0 < jobs[jj].f_end) return true;
        }
        return false;
    };
    vf32x4 raw[MEL_ROWS_NLD];
    auto fetch = [&](const MelJob &jb, uint32_t frame0) {  // rows past the end of the range are clamped here and never stored
        const gptr<const float> amp = as_global(jb.amp);
#pragma unroll
        for (int c = 0; c < MEL_ROWS_NLD; c++) {
            const uint32_t idx = 64u * c + lane, r = idx / MEL_ROWS_C4, c4 = idx - r * MEL_ROWS_C4;
            raw[c] = *reinterpret_cast<gptr<const vf32x4>>(amp + (size_t)min(frame0 + r, jb.f_end - 1) * amp_pitch + 4u * c4);
        }
    };
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    auto flush = [&](uint32_t mm_index) {
        if (minmax == nullptr) return;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lmin = fminf(lmin, __shfl_xor(lmin, o, 64));
            lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
        }
        if (lane == 0) {
            mel_atomic_min(&minmax[2 * mm_index], lmin);
            mel_atomic_max(&minmax[2 * mm_index + 1], lmax);
        }
        lmin = __builtin_inff();
        lmax = -__builtin_inff();
    };
    uint32_t frame0 = 0;
    bool have = locate(q, ji, frame0);
    if (have) fetch(jobs[ji], frame0);
    // the lane's filters: first bin (as an LDS float offset) and weights
    uint32_t lo[MAXG];
    float w[MAXG][W];
#pragma unroll
    for (int g = 0; g < MAXG; g++) {
        const uint32_t gg = (uint32_t)g < n_groups ? g : n_groups - 1u;
        lo[g] = tab[(gg * (W + 1)) * 64u + lane];
#pragma unroll
        for (int t = 0; t < W; t++) w[g][t] = __uint_as_float(tab[(gg * (W + 1) + 1 + t) * 64u + lane]);
    }
    while (have) {
        const MelJob job = jobs[ji];
        mel_wave_sync();  // the previous unit's reads are done
#pragma unroll
        for (int c = 0; c < MEL_ROWS_NLD; c++) {
            const uint32_t idx = 64u * c + lane, r = idx / MEL_ROWS_C4, c4 = idx - r * MEL_ROWS_C4;
            *reinterpret_cast<vf32x4 *>(tile + r * MEL_ROWS_AP + 4u * c4) = raw[c];
        }
        mel_wave_sync();
        uint32_t qn = q + 1, jn = ji, fn = 0;
        const bool have_next = locate(qn, jn, fn);
        if (have_next) fetch(jobs[jn], fn);
        const uint32_t rows = min(16u, job.f_end - frame0);
        gptr<float> orow = as_global(job.spec) + (size_t)frame0 * job.spec_pitch + lane;
        const float *row = tile;
        for (uint32_t r = 0; r < rows; r++, row += MEL_ROWS_AP, orow += job.spec_pitch) {
#pragma unroll
            for (int g = 0; g < MAXG; g++) {
                if ((uint32_t)g < n_groups) {  // wave-uniform
                    float a[W];
#pragma unroll
                    for (int t = 0; t < W; t++) a[t] = row[lo[g] + t];
                    float acc = 0.0f;
#pragma unroll
                    for (int t = 0; t < W; t++) acc = __builtin_fmaf(a[t], w[g][t], acc);
                    if (64u * g + lane < n_mel) {
                        const float d = 6.02059991327962390f * __builtin_amdgcn_logf(acc);  // dB_from_amp (decibel.rs:179-202)
                        orow[64 * g] = d;
                        lmin = fminf(lmin, d);
                        lmax = fmaxf(lmax, d);
                    }
                }
            }
        }
        if (!have_next || jn != ji) flush(job.mm_index);
        have = have_next;
        q = qn;
        ji = jn;
        frame0 = fn;
    }
}

hipError_t launch_mel_rows(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                           uint32_t amp_pitch, const uint32_t *d_tab, uint32_t n_groups, uint32_t n_mel, float *d_minmax,
                           uint32_t n_cu, hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    if (n_groups == 0 || n_groups > (uint32_t)MEL_ROWS_MAX_GROUPS || amp_pitch < (uint32_t)MEL_ROWS_COLS) return hipErrorInvalidValue;
    if (n_tiles > 0xffffffffu / (4u * MEL_MT)) return hipErrorInvalidValue;  // (the kernel counts 16-frame units in 32 bits)
    const uint32_t grid = n_tiles < 2u * n_cu ? n_tiles : 2u * n_cu;  // two workgroups per CU (LDS)
    hipLaunchKernelGGL(mel_rows_kernel<MEL_ROWS_W>, dim3(grid), dim3(256), 0, s, d_jobs, d_tile_start, n_jobs, n_tiles,
                       amp_pitch, d_tab, n_groups, n_mel, d_minmax);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Long rows (n_fft 4096: the 40 ms default at 88.2 / 96 kHz) under the default mel counts: mel_band_rows_kernel —
// the banded sums of the fused epilogues (mel_banded, stft_wave.h; table: build_mel_band, shared by the workgroup in LDS)
// over the amplitude rows the FFT kernel wrote.  A wave stages ONE row at a time in LDS (NLD coalesced 16-byte loads per
// lane, the next row requested before this row's sums start) and lane = mel: 136 taps per frame at the 96 kHz default
// against the matrix-core kernel's 16 x 16 x 4 products over blocks that are mostly zeros.  Persistent waves over
// contiguous frame ranges as in mel_rows_kernel.
// ------------------------------------------------------------------------------------------
struct MelBandArgs {
    uint32_t groups, words;
    uint32_t off[8], n[8];
};
template <int NLD>
__global__ __launch_bounds__(256) void mel_band_rows_kernel(const MelJob *__restrict__ jobs, const uint32_t *__restrict__ tile_start,
                                                            uint32_t n_jobs, uint32_t n_tiles, uint32_t amp_pitch,
                                                            const uint32_t *__restrict__ tab_g, MelBandArgs hb, uint32_t n_mel,
                                                            float *__restrict__ minmax) {
    constexpr uint32_t ROWP = 256u * NLD + 128u;  // staged floats + the zeros the widest group's taps may reach into
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    uint32_t *const tab = reinterpret_cast<uint32_t *>(smem_raw);
    float *const rows = reinterpret_cast<float *>(smem_raw) + ((hb.words + 3u) & ~3u);
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t i = tid; i < hb.words; i += 256) tab[i] = tab_g[i];
    float *const row = rows + wave * ROWP;
    row[256u * NLD + lane] = 0.0f;
    row[256u * NLD + 64u + lane] = 0.0f;
    __syncthreads();
    // a "unit" is one frame of the host's 128-frame tiles; every wave takes a contiguous range of units
    constexpr uint32_t UPT = MEL_TILE_FRAMES;
    const uint32_t n_units = n_tiles * UPT, n_w = gridDim.x * 4u, per = (n_units + n_w - 1) / n_w;
    const uint32_t q_end = min(n_units, (blockIdx.x * 4u + wave + 1u) * per);
    uint32_t q = (blockIdx.x * 4u + wave) * per;
    if (q >= q_end) return;
    uint32_t ji = mel_find_job(tile_start, n_jobs, q / UPT);
    auto locate = [&](uint32_t &qq, uint32_t &jj, uint32_t &frame) {
        while (qq < q_end) {
            const uint32_t t = qq / UPT;
            while (t >= tile_start[jj + 1]) jj++;
            frame = jobs[jj].f_begin + (t - tile_start[jj]) * UPT + qq % UPT;
            if (frame < jobs[jj].f_end) return true;
            qq = (t + 1u) * UPT;  // the rest of this tile is past the channel's last frame
        }
        return false;
    };
    vf32x4 raw[NLD];
    auto fetch = [&](const MelJob &jb, uint32_t frame) {
        const gptr<const float> a = as_global(jb.amp) + (size_t)frame * amp_pitch;
#pragma unroll
        for (int c = 0; c < NLD; c++) raw[c] = *reinterpret_cast<gptr<const vf32x4>>(a + min(256u * c + 4u * lane, amp_pitch - 4u));
    };
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    auto flush = [&](uint32_t mm_index) {
        if (minmax == nullptr) return;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lmin = fminf(lmin, __shfl_xor(lmin, o, 64));
            lmax = fmaxf(lmax, __shfl_xor(lmax, o, 64));
        }
        if (lane == 0) {
            mel_atomic_min(&minmax[2 * mm_index], lmin);
            mel_atomic_max(&minmax[2 * mm_index + 1], lmax);
        }
        lmin = __builtin_inff();
        lmax = -__builtin_inff();
    };
    uint32_t frame = 0;
    bool have = locate(q, ji, frame);
    if (have) fetch(jobs[ji], frame);
    while (have) {
        const MelJob job = jobs[ji];
        mel_wave_sync();  // the previous row's reads are done
#pragma unroll
        for (int c = 0; c < NLD; c++) *reinterpret_cast<vf32x4 *>(row + 256u * c + 4u * lane) = raw[c];
        mel_wave_sync();
        uint32_t qn = q + 1, jn = ji, fn = 0;
        const bool have_next = locate(qn, jn, fn);
        if (have_next) fetch(jobs[jn], fn);
        const gptr<float> orow = as_global(job.spec) + (size_t)frame * job.spec_pitch;
        mel_banded<TH_MEL_ROWS_PAIRED != 0>(lane, row, tab, hb.groups, hb.off, hb.n, [&](uint32_t m, float v) {
            if (m < n_mel) {
                const float d = 6.02059991327962390f * __builtin_amdgcn_logf(v);  // dB_from_amp (decibel.rs:179-202)
                orow[m] = d;
                lmin = fminf(lmin, d);
                lmax = fmaxf(lmax, d);
            }
        });
        if (!have_next || jn != ji) flush(job.mm_index);
        have = have_next;
        q = qn;
        ji = jn;
        frame = fn;
    }
}

template <int NLD>
static hipError_t launch_mel_band_rows_n(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                                         uint32_t amp_pitch, const uint32_t *d_tab, const MelBandArgs &hb, uint32_t n_mel,
                                         float *d_minmax, uint32_t n_cu, hipStream_t s) {
    auto kern = mel_band_rows_kernel<NLD>;
    const size_t lds = 4 * ((size_t)((hb.words + 3u) & ~3u) + 4 * (256 * (size_t)NLD + 128));
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const uint32_t per_cu = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, 160 * 1024 / lds));
    const uint32_t grid = n_tiles < per_cu * n_cu ? n_tiles : per_cu * n_cu;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, d_jobs, d_tile_start, n_jobs, n_tiles, amp_pitch, d_tab, hb, n_mel,
                       d_minmax);
    return hipGetLastError();
}
bool mel_band_rows_fits(uint32_t n_freq, uint32_t words) {
    const uint32_t nld = (n_freq + 255) / 256;
    // n_fft 4096.  (n_fft 8192, 17 loads per row: measured 7 % SLOWER than the matrix cores at the 192 kHz default — 18 KB rows
    // leave room for one workgroup of four waves per CU —, so those plans keep mel_mfma_kernel)
    if (nld != 9) return false;
    return 4 * ((size_t)((words + 3u) & ~3u) + 4 * (256 * (size_t)nld + 128)) <= 160 * 1024;
}
hipError_t launch_mel_band_rows(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                                uint32_t amp_pitch, uint32_t n_freq, const uint32_t *d_tab, uint32_t words, uint32_t groups,
                                const uint32_t *hdr, uint32_t n_mel, float *d_minmax, uint32_t n_cu, hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    if (n_tiles > 0xffffffffu / MEL_TILE_FRAMES || groups == 0 || groups > 8 || amp_pitch < 4) return hipErrorInvalidValue;
    MelBandArgs hb;
    hb.groups = groups;
    hb.words = words;
    for (int g = 0; g < 8; g++) {
        hb.off[g] = hdr[2 * g];
        hb.n[g] = hdr[2 * g + 1];
    }
    const uint32_t nld = (n_freq + 255) / 256;
    if (nld == 9) return launch_mel_band_rows_n<9>(d_jobs, d_tile_start, n_jobs, n_tiles, amp_pitch, d_tab, hb, n_mel, d_minmax, n_cu, s);
    return hipErrorInvalidValue;
}

hipError_t launch_mel_mfma(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                           uint32_t amp_pitch, const float *d_bt, const uint32_t *d_tile_band, const uint32_t *d_slice_start,
                           uint32_t n_slices, uint32_t zero_block, uint32_t n_mel, float *d_minmax, hipStream_t s) {
    if (!n_tiles || !n_slices) return hipSuccess;
    hipLaunchKernelGGL(mel_mfma_kernel, dim3(n_tiles, n_slices), dim3(256), 0, s, d_jobs, d_tile_start, n_jobs, amp_pitch,
                       d_bt, d_tile_band, d_slice_start, zero_block, n_mel, d_minmax);
    return hipGetLastError();
}

}  // namespace th
