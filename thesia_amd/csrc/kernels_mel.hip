// kernels_mel.hip — mel filterbank contraction on the CDNA4 matrix cores.
//
// Reference: `linspec.dot(&mel_fb)` (src-tauri/src/core/spectrogram.rs:207; ndarray → OpenBLAS
// sgemm), then dB_from_amp_inplace_default (:208, dynamics/decibel.rs:170-214):
//     mel[t, m] = sum_f |X|[t, f] * fb[f, m]      (T x F) . (F x n_mel),  f32
// This is the path's one genuine dense contraction.  It runs on `v_mfma_f32_16x16x4_f32` (f32 in,
// f32 accumulate: bit-for-bit an fmaf chain, no reduced-precision shortcut — gfx950 has no xf32).
//
// Shape per workgroup: 4 waves x 16 frames = 64 frames, all n_mel columns (NT tiles of 16 mels,
// accumulators in registers: 4*NT VGPRs).  K (= frequency bins) is walked in blocks of 16:
//   A operand: every lane loads ONE float4 of its frame row per K block (16 rows x 64 B per
//              wave-instruction); MFMA step s of the block uses element s, i.e. the block's 16 bins
//              are consumed in the order {s, 4+s, 8+s, 12+s} — a permutation of k, which a sum
//              does not care about — so no register shuffling is needed to feed the 16x16x4 shape.
//   B operand: filterbank values straight from L2 (the zero-padded table is <= 2 MB and shared by
//              every workgroup), one dword per lane per MFMA.
//   Band structure: a mel filter is a triangle, so for K block kb only the N tiles
//              [jlo[kb], jhi[kb]) hold non-zeros (≈ 1/7 of the dense product at 128 mels);
//              everything else is skipped with wave-uniform branches.
// Epilogue: 20*log10 via v_log_f32, store, min/max, one atomic pair per workgroup.
// Input is the linear amplitude written by stft_wave_kernel<..., AMP = true> (columns >= n_freq of
// the amplitude buffer are zero-filled once, so the padded K tail multiplies 0 * 0).
// HBM traffic per frame: 4*n_freq B read + 4*n_mel B written (the amplitude round trip through HBM
// is the price of not yet fusing this into the FFT kernel).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"

namespace th {

using f32x4 = __attribute__((ext_vector_type(4))) float;

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

template <int NT>
__global__ __launch_bounds__(256) void mel_mfma_kernel(const MelJob *__restrict__ jobs,
                                                       const uint32_t *__restrict__ tile_start, uint32_t n_jobs,
                                                       uint32_t n_kblocks, uint32_t amp_pitch,
                                                       const float *__restrict__ fb_pad,
                                                       const uint8_t *__restrict__ kb_jlo,
                                                       const uint8_t *__restrict__ kb_jhi, uint32_t n_mel,
                                                       float *__restrict__ minmax) {
    constexpr uint32_t NCOL = NT * 16;  // columns of the zero-padded filterbank
    __shared__ float red[8];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t ji = mel_find_job(tile_start, n_jobs, blockIdx.x);
    const MelJob job = jobs[ji];
    const uint32_t frame0 = job.f_begin + (blockIdx.x - tile_start[ji]) * MEL_TILE_FRAMES + wave * 16;
    const uint32_t kq = lane >> 4, li = lane & 15u;
    // rows past the end of the range are clamped for the loads and masked at the store
    const uint32_t my_row = min(frame0 + li, job.f_end - 1);
    const gptr<const float> arow = as_global(job.amp) + (size_t)my_row * amp_pitch + 4 * kq;
    const gptr<const float> fb = as_global(fb_pad) + (size_t)(4 * kq) * NCOL + li;

    f32x4 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; j++) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (frame0 < job.f_end) {  // wave-uniform: this wave has at least one frame
        for (uint32_t kb = 0; kb < n_kblocks; kb++) {
            const float4 a = *reinterpret_cast<gptr<const float4>>(arow + 16 * kb);
            const uint32_t jlo = kb_jlo[kb], jhi = kb_jhi[kb];
            const gptr<const float> bk = fb + (size_t)(16 * kb) * NCOL;
#pragma unroll
            for (int j = 0; j < NT; j++) {
                if ((uint32_t)j >= jlo && (uint32_t)j < jhi) {  // wave-uniform band test
                    const gptr<const float> bp = bk + 16 * j;
                    const float b0 = bp[0], b1 = bp[NCOL], b2 = bp[2 * NCOL], b3 = bp[3 * NCOL];
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b0, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b1, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b2, acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b3, acc[j], 0, 0, 0);
                }
            }
        }
    }

    // C/D layout of the 16x16 shapes: col = lane & 15, row = 4 * (lane >> 4) + r
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    const gptr<float> spec = as_global(job.spec);
#pragma unroll
    for (int j = 0; j < NT; j++) {
        const uint32_t m = 16 * j + li;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const uint32_t f = frame0 + 4 * kq + r;
            if (f < job.f_end && m < n_mel) {
                // dB_from_amp (decibel.rs:179-202): 20*log10(x); x = +0 -> -inf
                const float d = 6.02059991327962390f * __builtin_amdgcn_logf(acc[j][r]);
                spec[(size_t)f * job.spec_pitch + m] = d;
                lmin = fminf(lmin, d);
                lmax = fmaxf(lmax, d);
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

hipError_t launch_mel_mfma(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                           uint32_t n_kblocks, uint32_t amp_pitch, const float *d_fb_pad, uint32_t n_ntiles,
                           const uint8_t *d_kb_jlo, const uint8_t *d_kb_jhi, uint32_t n_mel, float *d_minmax,
                           hipStream_t s) {
    if (!n_tiles) return hipSuccess;
#define TH_MEL_CASE(NT)                                                                                             \
    if (n_ntiles <= (NT)) {                                                                                         \
        hipLaunchKernelGGL(mel_mfma_kernel<NT>, dim3(n_tiles), dim3(256), 0, s, d_jobs, d_tile_start, n_jobs, n_kblocks, \
                           amp_pitch, d_fb_pad, d_kb_jlo, d_kb_jhi, n_mel, d_minmax);                               \
        return hipGetLastError();                                                                                   \
    }
    TH_MEL_CASE(8)
    TH_MEL_CASE(16)
    TH_MEL_CASE(24)
    TH_MEL_CASE(32)
#undef TH_MEL_CASE
    return hipErrorInvalidValue;
}

}  // namespace th
