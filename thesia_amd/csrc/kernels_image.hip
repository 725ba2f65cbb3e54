// kernels_image.hip — f32 dB spec → u16 grey image (transposing quantise) and the level-0
// colormap raster.  Integer / rounding stages are bit-exact restatements of the reference:
//   convert_spectrogram_to_img   src-tauri/src/core/visualize/drawing.rs:4-33
//   colour index + RGBA write    src-tauri/src/core/render_tiles.rs:339-351
// Both kernels are pure HBM streaming (read 4 B + write 2 B per pixel; read 2 B + write 4 B).
#include <hip/hip_runtime.h>

#include "kernels.h"
#include "stft_core.h"

// Integer / rounding stages must round once per operation exactly like the reference's scalar f32
// code: no FMA contraction anywhere in this file (hipcc defaults to -ffp-contract=fast; the
// __fmul_rn/__fadd_rn header helpers are plain operators compiled under that default, so they
// still fuse after inlining — use plain operators below this pragma instead; the build also
// passes -ffp-contract=off for this file).
#pragma clang fp contract(off)

namespace th {

__device__ __forceinline__ uint32_t find_job(const uint32_t *__restrict__ start, uint32_t n, uint32_t b) {
    uint32_t lo = 0, hi = n;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (start[mid] <= b) lo = mid;
        else hi = mid;
    }
    return lo;
}

// drawing.rs:26-28 in f32, one rounding per operation (no FMA contraction):
//   zero_to_one = (dB - min) / span;  u = zero_to_one * u16_span + min_value;
//   u.round().clamp(0, 65535) as u16   (NaN -> 0)
__device__ __forceinline__ uint16_t quantise(float dB, float min_dB, float span, float u16_span, float min_value) {
    const float z = (dB - min_dB) / span;          // plain operators: contraction is off in this file
    const float u = z * u16_span + min_value;
    const float r = roundf(u);  // half away from zero, like f32::round
    if (__builtin_isnan(r)) return 0;
    return (uint16_t)fminf(fmaxf(r, 0.0f), 65535.0f);
}

// One block = one 64(frames) x 64(freq rows) tile: coalesced f32 row reads (frame-major spec),
// quantise, transpose through LDS, coalesced u16 row writes (freq-major image).
__global__ __launch_bounds__(256) void spec_to_img_kernel(const ImgJob *__restrict__ jobs,
                                                          const uint32_t *__restrict__ tile_start, uint32_t n_jobs,
                                                          float min_dB, float span, float u16_span,
                                                          float min_value) {
    __shared__ uint16_t tile[IMG_TILE][IMG_TILE + 2];  // [freq][frame], +2 keeps rows 4-byte aligned & spreads banks
    const uint32_t ji = find_job(tile_start, n_jobs, blockIdx.x);
    const ImgJob job = jobs[ji];
    const uint32_t local = blockIdx.x - tile_start[ji];
    const uint32_t tiles_t = (job.n_frames + IMG_TILE - 1) / IMG_TILE;
    const uint32_t t0 = (local % tiles_t) * IMG_TILE;
    const uint32_t r0 = (local / tiles_t) * IMG_TILE;  // image row (relative to i_start)
    const uint32_t out_h = job.i_end - job.i_start;
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;

    // read: lanes along frequency (contiguous in the spec)
    const uint32_t i_freq = job.i_start + r0 + tx;
#pragma unroll 4
    for (uint32_t dt = ty; dt < IMG_TILE; dt += 4) {
        const uint32_t t = t0 + dt;
        uint16_t px = 0;
        if (t < job.n_frames && i_freq < job.height && r0 + tx < out_h)
            px = quantise(as_global(job.spec)[(size_t)t * job.height + i_freq], min_dB, span, u16_span, min_value);
        tile[tx][dt] = px;
    }
    __syncthreads();
    // write: lanes along time (contiguous in the image)
#pragma unroll 4
    for (uint32_t dr = ty; dr < IMG_TILE; dr += 4) {
        const uint32_t r = r0 + dr, t = t0 + tx;
        if (r < out_h && t < job.n_frames) as_global(job.img)[(size_t)r * job.n_frames + t] = tile[dr][tx];
    }
}

hipError_t launch_spec_to_img(const ImgJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs,
                              uint32_t n_tiles, float min_dB, float max_dB, uint32_t colormap_len, hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    // drawing.rs:20-22 — min_value = max(round(65535 / C), 1) in f64; u16_span = (65535 - min_value) as f32
    uint32_t min_value = 1;
    if (colormap_len) {
        const double r = __builtin_round(65535.0 / (double)colormap_len);
        const uint32_t v = r >= 65535.0 ? 65535u : (uint32_t)r;
        min_value = v > 1 ? v : 1;
    }
    const float span = max_dB - min_dB;
    hipLaunchKernelGGL(spec_to_img_kernel, dim3(n_tiles), dim3(256), 0, s, d_jobs, d_tile_start, n_jobs, min_dB, span,
                       (float)(65535u - min_value), (float)min_value);
    return hipGetLastError();
}

// Level-0 raster: out row r of a tile = image row (origin_y + height - 1 - r) (render_tiles.rs:340),
// colour index = (v * (C - 1) + 32767) / 65535 in integer arithmetic (:342-346), RGBA from the LUT.
__global__ __launch_bounds__(256) void raster_level0_kernel(const RasterJob *__restrict__ jobs,
                                                            const uint32_t *__restrict__ block_start,
                                                            uint32_t n_jobs, const uint32_t *__restrict__ colormap,
                                                            uint32_t n_colors) {
    __shared__ uint32_t lut[1024];
    const bool use_lds = n_colors <= 1024;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < n_colors; i += 256) lut[i] = colormap[i];
        __syncthreads();
    }
    const uint32_t ji = find_job(block_start, n_jobs, blockIdx.x);
    const RasterJob job = jobs[ji];
    const uint32_t n_px = job.width * job.height;
    const uint32_t base = (blockIdx.x - block_start[ji]) * RASTER_PIXELS_PER_BLOCK;
    const gptr<uint32_t> out = as_global(reinterpret_cast<uint32_t *>(job.rgba));
#pragma unroll
    for (uint32_t it = 0; it < RASTER_PIXELS_PER_BLOCK / 256; it++) {
        const uint32_t p = base + it * 256 + threadIdx.x;
        if (p >= n_px) break;
        const uint32_t r = p / job.width, c = p - r * job.width;
        const uint32_t src_row = job.origin_y + (job.height - 1 - r);
        const uint32_t v = as_global(job.img)[(size_t)src_row * job.img_width + job.origin_x + c];
        const uint32_t ci = n_colors <= 1 ? 0 : (v * (n_colors - 1) + 32767u) / 65535u;
        uint32_t rgba;  // two explicit loads: a select between an LDS and a global pointer would go flat
        if (use_lds) rgba = lut[ci];
        else rgba = colormap[ci];
        out[p] = rgba;
    }
}

hipError_t launch_raster_level0(const RasterJob *d_jobs, const uint32_t *d_block_start, uint32_t n_jobs,
                                uint32_t n_blocks, const uint8_t *d_colormap, uint32_t n_colors, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    hipLaunchKernelGGL(raster_level0_kernel, dim3(n_blocks), dim3(256), 0, s, d_jobs, d_block_start, n_jobs,
                       reinterpret_cast<const uint32_t *>(d_colormap), n_colors);
    return hipGetLastError();
}

}  // namespace th
