// kernels_image.hip — f32 dB spec → u16 grey image (transposing quantise) and the level-0
// colormap raster.  Integer / rounding stages are bit-exact restatements of the reference:
//   convert_spectrogram_to_img   src-tauri/src/core/visualize/drawing.rs:4-33
//   colour index + RGBA write    src-tauri/src/core/render_tiles.rs:339-351
// Both kernels are pure HBM streaming (read 4 B + write 2 B per pixel; read 2 B + write 4 B).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "kernels.h"
#include "stft_core.h"

// Integer / rounding stages must round once per operation exactly like the reference's scalar f32
// code: no FMA contraction anywhere in this file (hipcc defaults to -ffp-contract=fast; the
// __fmul_rn/__fadd_rn header helpers are plain operators compiled under that default, so they
// still fuse after inlining — use plain operators below this pragma instead; the build also
// passes -ffp-contract=off for this file).
#pragma clang fp contract(off)

namespace th {

// Block -> job: a host-built table with one entry per block (one scalar load), not a binary search over the
// per-job prefix sums: the search is a chain of ~log2(n_jobs) dependent loads (microseconds) in front of
// every block's first data request, during which the block holds its LDS and wave slots for nothing.

// drawing.rs:26-28 in f32, one rounding per operation (no FMA contraction):
//   zero_to_one = (dB - min) / span;  u = zero_to_one * u16_span + min_value;
//   u.round().clamp(0, 65535) as u16   (NaN -> 0)
// The division by the launch-uniform span is the correctly rounded quotient obtained from the correctly rounded
// reciprocal with one FMA correction step (Markstein): q0 = a*r, e = fma(-q0, span, a), q = fma(e, r, q0) — 3
// operations instead of the ~12 of the generic IEEE division sequence (this kernel ran 36 VALU instructions per
// pixel).  Verified bit for bit against a / span for all 2^32 dividends and 60 divisors spread over [1e-3, 2e3]
// (tests/test_host_logic.py keeps a sampled version); non-finite dividends bypass it, spans outside the
// verified range take the plain division (rinv = 0).
template <bool FAST>
__device__ __forceinline__ uint32_t quantise(float dB, float min_dB, float span, float rinv, float u16_span, float min_value) {
    const float a = dB - min_dB;  // plain operators: contraction is off in this file
    float z;
    if constexpr (FAST) {
        const float q0 = a * rinv;
        const float e = __builtin_fmaf(-q0, span, a);
        const float q = __builtin_fmaf(e, rinv, q0);
        z = __builtin_fabsf(a) < __builtin_inff() ? q : a;  // +-inf / NaN: a / span == a for a finite span > 0
    } else {
        z = a / span;
    }
    const float u = z * u16_span + min_value;
    // f32::round (half away from zero) followed by clamp(0, 65535): for u >= 0 that is trunc(u) + (frac >= 0.5);
    // for u < 0 any value <= 0 clamps to 0, and trunc(u) + 0 is one
    const float t = __builtin_truncf(u);
    const float r = t + ((u - t) >= 0.5f ? 1.0f : 0.0f);
    return (uint32_t)fminf(fmaxf(r, 0.0f), 65535.0f);  // fmaxf(NaN, 0) = 0: NaN -> 0 like Rust's saturating `as u16`
}
// reciprocal for quantise(): non-zero only inside the range the FMA-corrected quotient was verified for
__host__ __device__ __forceinline__ float quantise_rinv(float span) {
    return (span >= 1e-3f && span <= 2e3f) ? 1.0f / span : 0.0f;
}

// One block = one tile of IMG_TILE_F = 128 frequency rows x IMG_TILE_T = 64 frames.  Read: lanes along frequency
// (contiguous in the frame-major spec), two bins per lane: a wave-instruction fetches 512 contiguous bytes of one
// frame row.  Quantise, transpose through a u16 LDS tile.  Write: lanes along time, two frames per lane as one dword:
// the 64 frames of an image row are one 128-byte line, a wave-instruction writes two rows.
// Shape chosen by measurement (scripts/ubench/transpose_shapes.hip, same traffic without the arithmetic):
// 128 x 64 streams at 5.3-5.5 TB/s, the earlier 64 freq x 128 frames tile at 3.9-4.1, larger tiles worse.
// LDS pitch 66 u16 = 33 dwords (odd): the row-wise ds_read_b32 of the write phase are conflict-free, the
// transposing ds_write_b16 of the read phase two-way at worst.
constexpr uint32_t IMG_LDS_PITCH = IMG_TILE_T + 2;

__global__ __launch_bounds__(256) void spec_to_img_kernel(const ImgJob *__restrict__ jobs,
                                                          const uint32_t *__restrict__ tile_job, uint32_t n_jobs,
                                                          float min_dB, float span, float u16_span,
                                                          float min_value, const float *__restrict__ d_range) {
    __shared__ __attribute__((aligned(4))) uint16_t tile[IMG_TILE_F][IMG_LDS_PITCH];  // [freq][frame]
    bool all_zero = false;
    float rinv = quantise_rinv(span);
    if (d_range != nullptr) {  // (min_dB, max_dB) left on the device by th_global_db_range_dev
        const float lo = d_range[0], hi = d_range[1];
        min_dB = lo;
        span = hi - lo;
        rinv = quantise_rinv(span);
        all_zero = lo == hi && __builtin_isinf(hi) && hi < 0.0f;  // every value -inf: zero image (drawing.rs:16-18)
    }
    const ImgJob job = jobs[tile_job[blockIdx.x]];
    const gptr<const float> spec = as_global(job.spec);
    const gptr<uint16_t> img = as_global(job.img);
    const uint32_t out_h = job.i_end - job.i_start;
    // Tile order: frequency-tile fastest, and consecutive tiles on the SAME XCD (workgroups are dealt round-robin
    // over the 8 XCDs, so blocks b and b+8 share an L2): frequency-adjacent tiles read adjacent 512-byte segments of
    // the same spec rows, so whatever straddles a line boundary (unpadded row pitches) is found in that L2.
    const uint32_t n_local = job.n_tiles;
    const uint32_t local0 = blockIdx.x - job.first_tile;
    const uint32_t per_xcd = (n_local + 7) / 8;
    uint32_t local = (local0 % 8) * per_xcd + local0 / 8;  // bijective when n_local % 8 == 0
    if (n_local % 8 != 0) local = local0;                   // ragged tail: plain order
    const uint32_t tiles_r = (out_h + IMG_TILE_F - 1) / IMG_TILE_F;
    const uint32_t r0 = (local % tiles_r) * IMG_TILE_F;  // image row (relative to i_start)
    const uint32_t t0 = (local / tiles_r) * IMG_TILE_T;
    const uint32_t lane = threadIdx.x & 63;
    // wave-uniform (tell the compiler): row pointers and the per-row predicates then live in SGPRs, and every access is
    // "scalar row base + 32-bit lane offset" with no per-element 64-bit VALU address arithmetic
    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);

    // read: lanes along frequency (contiguous in the spec), bins 2*lane and 2*lane + 1 of the tile
    const uint32_t i_freq = job.i_start + r0 + 2 * lane;
    const bool ok0 = i_freq < job.height && r0 + 2 * lane < out_h;
    const bool ok1 = i_freq + 1 < job.height && r0 + 2 * lane + 1 < out_h;
    // 8-byte loads need an even element offset (even row pitch, even first bin, 8-byte aligned base) and one float of
    // row padding, because the pair of the last bin reads one element past `height`: all launch-uniform.
    const bool al2 = (reinterpret_cast<uintptr_t>(job.spec) & 7u) == 0 && job.spec_pitch % 2 == 0 && (job.i_start + r0) % 2 == 0 &&
                     job.spec_pitch > job.height;
    // all 16 row segments of this wave are requested before the first is consumed: one HBM latency per tile.  The loads
    // are unconditional (indices clamped, results discarded by selects): a per-lane branch around a load makes the
    // compiler wait for the previous one.
    constexpr uint32_t NLD = IMG_TILE_T / 4;
    float v0[NLD], v1[NLD];
    const float qnan = __builtin_nanf("");
    const uint32_t f0c = min(i_freq, (job.height - 1) & ~1u), f1c = min(i_freq + 1, job.height - 1);  // height >= 1 here
    if (al2) {
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) {
            const uint32_t t = t0 + wv + 4 * i;
            const gptr<const float> rowp = spec + (size_t)min(t, job.n_frames - 1) * job.spec_pitch;  // scalar
            const float2 x = *reinterpret_cast<gptr<const float2>>(rowp + f0c);
            v0[i] = (ok0 && t < job.n_frames) ? x.x : qnan;
            v1[i] = (ok1 && t < job.n_frames) ? x.y : qnan;
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) {
            const uint32_t t = t0 + wv + 4 * i;
            const gptr<const float> rowp = spec + (size_t)min(t, job.n_frames - 1) * job.spec_pitch;  // scalar
            const float a = rowp[min(i_freq, job.height - 1)], b2 = rowp[f1c];
            v0[i] = (ok0 && t < job.n_frames) ? a : qnan;
            v1[i] = (ok1 && t < job.n_frames) ? b2 : qnan;
        }
    }
    // the three cases are launch-uniform: one specialised loop each instead of a branch per element
    if (all_zero) {
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) tile[2 * lane][wv + 4 * i] = tile[2 * lane + 1][wv + 4 * i] = 0;
    } else if (rinv != 0.0f) {
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) {
            tile[2 * lane][wv + 4 * i] = (uint16_t)quantise<true>(v0[i], min_dB, span, rinv, u16_span, min_value);  // NaN -> 0
            tile[2 * lane + 1][wv + 4 * i] = (uint16_t)quantise<true>(v1[i], min_dB, span, rinv, u16_span, min_value);
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < NLD; i++) {
            tile[2 * lane][wv + 4 * i] = (uint16_t)quantise<false>(v0[i], min_dB, span, rinv, u16_span, min_value);
            tile[2 * lane + 1][wv + 4 * i] = (uint16_t)quantise<false>(v1[i], min_dB, span, rinv, u16_span, min_value);
        }
    }
    __syncthreads();
    // write: lanes along time, two frames per lane; a wave-instruction covers two image rows (lanes 0-31 / 32-63).
    // All LDS reads first (they are unconditional), then the stores: one LDS latency per wave instead of one per row.
    constexpr uint32_t NST = IMG_TILE_F / 8;  // 2 rows per instruction, 4 waves
    const uint32_t half = lane >> 5, tl = 2 * (lane & 31u);
    const uint32_t t = t0 + tl;
    uint32_t pairs[NST];
#pragma unroll
    for (uint32_t i = 0; i < NST; i++) pairs[i] = *reinterpret_cast<const uint32_t *>(&tile[2 * (wv + 4 * i) + half][tl]);
    // (t is even; a row base is 4-byte aligned when the image base is and the pitch is even: launch-uniform)
    const bool rows_al = (reinterpret_cast<uintptr_t>(job.img) & 3u) == 0 && job.img_pitch % 2 == 0;
    // Rows at the library's padded pitch (th_pitch_u16): the padding is ours, and the tile holds zeros there (NaN -> 0), so
    // store the whole last 128-byte line of every row — a partially written line costs HBM a read-modify-write.
    const uint32_t t_lim = (job.img_pitch % IMG_TILE_T == 0 && job.img_pitch - job.n_frames < IMG_TILE_T) ? job.img_pitch : job.n_frames;
#pragma unroll
    for (uint32_t i = 0; i < NST; i++) {
        const uint32_t r = r0 + 2 * (wv + 4 * i) + half;
        const uint32_t pair = pairs[i];
        const gptr<uint16_t> rowo = img + (size_t)r * job.img_pitch;
        if (r < out_h) {
            if (t + 1 < t_lim && rows_al) {
                *reinterpret_cast<gptr<uint32_t>>(rowo + t) = pair;
            } else {
                if (t < t_lim) rowo[t] = (uint16_t)(pair & 0xffffu);
                if (t + 1 < t_lim) rowo[t + 1] = (uint16_t)(pair >> 16);
            }
        }
    }
}

// update_spec_imgs' range clamp (core/mod.rs:179-180) on the device: in = [min, -max] as th_minmax_reduce_dev (and a
// MIN all-reduce across ranks) leaves it, out = [min_dB, max_dB].
__global__ void db_range_kernel(const float *__restrict__ in, float dB_range, float *__restrict__ out) {
    float mx = -in[1];
    mx = fminf(mx, 0.0f);
    const float mn = fmaxf(in[0], mx - dB_range);
    out[0] = mn;
    out[1] = mx;
}
hipError_t launch_db_range(const float *d_in, float dB_range, float *d_out, hipStream_t s) {
    hipLaunchKernelGGL(db_range_kernel, dim3(1), dim3(1), 0, s, d_in, dB_range, d_out);
    return hipGetLastError();
}

hipError_t launch_spec_to_img(const ImgJob *d_jobs, const uint32_t *d_tile_job, uint32_t n_jobs,
                              uint32_t n_tiles, float min_dB, float max_dB, uint32_t colormap_len, const float *d_range,
                              hipStream_t s) {
    if (!n_tiles) return hipSuccess;
    // drawing.rs:20-22 — min_value = max(round(65535 / C), 1) in f64; u16_span = (65535 - min_value) as f32
    uint32_t min_value = 1;
    if (colormap_len) {
        const double r = __builtin_round(65535.0 / (double)colormap_len);
        const uint32_t v = r >= 65535.0 ? 65535u : (uint32_t)r;
        min_value = v > 1 ? v : 1;
    }
    const float span = max_dB - min_dB;
    hipLaunchKernelGGL(spec_to_img_kernel, dim3(n_tiles), dim3(256), 0, s, d_jobs, d_tile_job, n_jobs, min_dB, span,
                       (float)(65535u - min_value), (float)min_value, d_range);
    return hipGetLastError();
}

// Level-0 raster: out row r of a tile = image row (origin_y + height - 1 - r) (render_tiles.rs:340),
// colour index = (v * (C - 1) + 32767) / 65535 in integer arithmetic (:342-346), RGBA from the LUT.
// A thread rasterises quads of 4 horizontally adjacent pixels: 4 u16 loads, one 16-byte store
// (1 KiB per wave-instruction) when the tile row pitch allows it.
// x / 65535 == (x + 1 + (x >> 16)) >> 16 for every x < 2^32 - 65536 (checked exhaustively), i.e. for every
// v <= 65535 and n_colors <= 65536 (the host entry points require that): three full-rate operations instead of a
// quarter-rate 32-bit multiply-high.
// v <= 65535 and n_colors - 1 <= 65535 fit v_mul_u32_u24 / v_mad_u32_u24 (full rate; the 32-bit v_mul_lo_u32 the plain
// product compiles to is a quarter-rate instruction).  One colour (or none): n_colors - 1 = 0 gives x = 32767 -> index 0, the
// reference's `C <= 1 -> 0` (render_tiles.rs:342-343), without a branch in front of every pixel.
__device__ __forceinline__ uint32_t colour_index(uint32_t v, uint32_t n_colors) {
    const uint32_t cm1 = n_colors > 1u ? n_colors - 1u : 0u;  // launch-uniform: a scalar
    const uint32_t x = __umul24(v, cm1) + 32767u;
    return (x + 1u + (x >> 16)) >> 16;
}

// LUT_IN_LDS is a template parameter on purpose: a run-time select between an LDS and a global
// LUT pointer would make the load a flat_load.
template <bool LUT_IN_LDS>
__device__ __forceinline__ void raster_quads(const RasterJob &job, uint32_t base, const uint32_t *lut,
                                             gptr<const uint32_t> colormap, uint32_t n_colors) {
    const gptr<const uint16_t> img = as_global(job.img);
    const gptr<uint32_t> out = as_global(reinterpret_cast<uint32_t *>(job.rgba));
    const bool base_aligned = (reinterpret_cast<uintptr_t>(job.rgba) & 15u) == 0;
    // 8-byte aligned source quads: base pointer, row pitch and tile origin all multiples of 4 pixels
    const bool src_al = (reinterpret_cast<uintptr_t>(job.img) & 7u) == 0 && job.img_pitch % 4 == 0 && job.origin_x % 4 == 0;
    auto look = [&](uint32_t v) -> uint32_t {
        const uint32_t ci = colour_index(v, n_colors);
        if constexpr (LUT_IN_LDS) return lut[ci];
        else return colormap[ci];
    };
    if (job.width % 4 == 0 && base_aligned) {
        // row quads: 4 horizontally adjacent pixels of one tile row per thread
        const uint32_t n_quads = job.quads_per_row * job.height;
        constexpr uint32_t NIT = RASTER_QUADS_PER_BLOCK / RASTER_THREADS;
        if (src_al && base_aligned && base + RASTER_QUADS_PER_BLOCK <= n_quads) {
            // fast path (block-uniform): the whole block is inside the tile and everything is aligned — request
            // all NIT source quads of the thread first (NIT x 8 bytes in flight per lane instead of one), then
            // look up and store
            uint2 w[NIT];
            uint32_t o[NIT];
#pragma unroll
            for (uint32_t it = 0; it < NIT; it++) {
                const uint32_t q = base + it * RASTER_THREADS + threadIdx.x;
                const uint32_t r = job.quads_per_row == 1 ? q : __umulhi(q, job.inv_qpr);
                const uint32_t c = (q - r * job.quads_per_row) * 4;
                const uint32_t src_row = job.origin_y + (job.height - 1 - r);
                w[it] = *reinterpret_cast<gptr<const uint2>>(img + ((size_t)src_row * job.img_pitch + job.origin_x + c));
                o[it] = r * job.width + c;
            }
#pragma unroll
            for (uint32_t it = 0; it < NIT; it++) {
                *reinterpret_cast<gptr<uint4>>(out + o[it]) =
                    make_uint4(look(w[it].x & 0xffffu), look(w[it].x >> 16), look(w[it].y & 0xffffu), look(w[it].y >> 16));
            }
            return;
        }
#pragma unroll 4
        for (uint32_t it = 0; it < NIT; it++) {
            const uint32_t q = base + it * RASTER_THREADS + threadIdx.x;
            if (q >= n_quads) break;
            const uint32_t r = job.quads_per_row == 1 ? q : __umulhi(q, job.inv_qpr);  // 2^32/1 does not fit inv_qpr
            const uint32_t c = (q - r * job.quads_per_row) * 4;
            const uint32_t src_row = job.origin_y + (job.height - 1 - r);  // first output row = highest frequency
            const gptr<const uint16_t> src = img + ((size_t)src_row * job.img_pitch + job.origin_x + c);
            const uint32_t o = r * job.width + c;
            uint32_t v0, v1, v2, v3;
            if (src_al) {  // 8-byte aligned quads (uniform per job): one load instead of four 2-byte loads
                const uint2 w = *reinterpret_cast<gptr<const uint2>>(src);
                v0 = w.x & 0xffffu;
                v1 = w.x >> 16;
                v2 = w.y & 0xffffu;
                v3 = w.y >> 16;
            } else {
                v0 = src[0];
                v1 = src[1];
                v2 = src[2];
                v3 = src[3];
            }
            const uint32_t p0 = look(v0), p1 = look(v1), p2 = look(v2), p3 = look(v3);
            if (base_aligned) {
                *reinterpret_cast<gptr<uint4>>(out + o) = make_uint4(p0, p1, p2, p3);
            } else {
                out[o] = p0;
                out[o + 1] = p1;
                out[o + 2] = p2;
                out[o + 3] = p3;
            }
        }
    } else {
        // flat quads (tile widths that are not multiples of 4: the last tile column of most images): the tile's RGBA
        // output is one contiguous array of height*width pixels and a thread owns 4 consecutive OUTPUT pixels, so the
        // stores stay 16 bytes wide and 16-byte aligned whatever the width.  The 4 pixels are consecutive in the source
        // row too, except that a quad may run over the end of a row: output row r + 1 is image row - 1, i.e. the pixels
        // past the end sit at the same offsets minus (pitch + width).  Branch-free: one select per pixel.
        // (Row quads with 4-byte-aligned 16-byte stores were 40 % slower per pixel; the first version of this path
        // recomputed row and column per pixel with 64-bit multiplies and cost 23 % of the whole kernel for 9 % of the
        // bench's pixels.)
        // A tile base that is only 4-byte aligned (tiles packed back to back) is handled here too: quads are laid on the
        // 16-byte grid of the ADDRESSES, i.e. quad q covers output pixels 4q - mis .. 4q - mis + 3 with mis = the base's
        // offset from the grid in pixels; only the first and the last quad of the tile are partial.
        const uint32_t n_px = job.width * job.height;
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(job.rgba) >> 2) & 3u;
        const uint32_t n_quads = (n_px + mis + 3) / 4;
        const int32_t wrap = -(int32_t)(job.img_pitch + job.width);
#pragma unroll 4
        for (uint32_t it = 0; it < RASTER_QUADS_PER_BLOCK / RASTER_THREADS; it++) {
            const uint32_t q = base + it * RASTER_THREADS + threadIdx.x;
            if (q >= n_quads) break;
            const int32_t o = (int32_t)(4 * q) - (int32_t)mis;  // first output pixel of the quad (negative: before the tile)
            const uint32_t o0 = o < 0 ? 0u : (uint32_t)o;
            const uint32_t r = job.width == 1 ? o0 : __umulhi(o0, job.inv_width);  // o0 / width
            const uint32_t c = o0 - r * job.width;
            if (o >= 0 && (uint32_t)o + 4 <= n_px && job.width >= 4) {
                const gptr<const uint16_t> src = img + ((size_t)(job.origin_y + (job.height - 1 - r)) * job.img_pitch + job.origin_x + c);
                uint32_t px[4];
#pragma unroll
                for (int i = 0; i < 4; i++) px[i] = look(src[(int32_t)i + (c + i >= job.width ? wrap : 0)]);
                *reinterpret_cast<gptr<uint4>>(out + o) = make_uint4(px[0], px[1], px[2], px[3]);
            } else {  // the first / last quad of the tile, tiles narrower than 4 pixels: pixel by pixel
                uint32_t rr = r, cc = c;
                for (int32_t p = (int32_t)o0; p < o + 4 && (uint32_t)p < n_px; p++) {
                    out[p] = look(img[(size_t)(job.origin_y + (job.height - 1 - rr)) * job.img_pitch + job.origin_x + cc]);
                    if (++cc == job.width) {
                        cc = 0;
                        rr++;
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(RASTER_THREADS) void raster_level0_kernel(const RasterJob *__restrict__ jobs,
                                                            const uint32_t *__restrict__ block_job,
                                                            uint32_t n_chunks, const uint32_t *__restrict__ colormap,
                                                            uint32_t n_colors) {
    __shared__ uint32_t lut[1024];
    const bool use_lds = n_colors <= 1024;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < n_colors; i += RASTER_THREADS) lut[i] = colormap[i];
        __syncthreads();
    }
    // chunk loop: with a grid smaller than the number of chunks (launch_raster_level0) a block keeps its LUT and
    // walks chunks b, b + gridDim.x, ...
    for (uint32_t b = blockIdx.x; b < n_chunks; b += gridDim.x) {
        const RasterJob job = jobs[block_job[b]];
        const uint32_t base = (b - job.first_block) * RASTER_QUADS_PER_BLOCK;
        if (use_lds) raster_quads<true>(job, base, lut, as_global(colormap), n_colors);
        else raster_quads<false>(job, base, lut, as_global(colormap), n_colors);
    }
}

// One tile, job passed by value (the tile-request path of the TrackManager: no descriptor tables to upload)
__global__ __launch_bounds__(RASTER_THREADS) void raster_tile_kernel(RasterJob job, const uint32_t *__restrict__ colormap,
                                                                      uint32_t n_colors) {
    __shared__ uint32_t lut[1024];
    const bool use_lds = n_colors <= 1024;
    if (use_lds) {
        for (uint32_t i = threadIdx.x; i < n_colors; i += RASTER_THREADS) lut[i] = colormap[i];
        __syncthreads();
    }
    const uint32_t base = blockIdx.x * RASTER_QUADS_PER_BLOCK;
    if (use_lds) raster_quads<true>(job, base, lut, as_global(colormap), n_colors);
    else raster_quads<false>(job, base, lut, as_global(colormap), n_colors);
}

hipError_t launch_raster_tile(const uint16_t *d_img, uint32_t img_width, uint32_t img_height, uint32_t img_pitch,
                              uint32_t origin_x, uint32_t origin_y, uint32_t width, uint32_t height, uint8_t *d_rgba,
                              const uint8_t *d_colormap, uint32_t n_colors, hipStream_t s) {
    if (!width || !height) return hipSuccess;
    const uint32_t qpr = (width + 3) / 4;
    const uint32_t inv = qpr > 1 ? (uint32_t)((1ull << 32) / qpr) + 1u : 0u;
    const uint32_t inv_w = width > 1 ? (uint32_t)((1ull << 32) / width) + 1u : 0u;
    const uint32_t nb = (uint32_t)(((uint64_t)qpr * height + 1 + RASTER_QUADS_PER_BLOCK - 1) / RASTER_QUADS_PER_BLOCK);
    const RasterJob job{d_img, d_rgba, img_width, img_height, origin_x, origin_y, width, height, img_pitch, qpr, inv, inv_w, 0u};
    hipLaunchKernelGGL(raster_tile_kernel, dim3(nb), dim3(RASTER_THREADS), 0, s, job,
                       reinterpret_cast<const uint32_t *>(d_colormap), n_colors);
    return hipGetLastError();
}

hipError_t launch_raster_level0(const RasterJob *d_jobs, const uint32_t *d_block_job, uint32_t n_jobs,
                                uint32_t n_blocks, const uint8_t *d_colormap, uint32_t n_colors, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    const uint32_t grid = n_blocks;
    hipLaunchKernelGGL(raster_level0_kernel, dim3(grid), dim3(RASTER_THREADS), 0, s, d_jobs, d_block_job, n_blocks,
                       reinterpret_cast<const uint32_t *>(d_colormap), n_colors);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Quantise + level-0 raster in ONE pass over the f32 spec (round 4; drawing.rs:4-33 + render_tiles.rs:290-351): 10 bytes per
// pixel (4 read, 2 + 4 written) where spec_to_img_kernel + raster_level0_kernel move 6 + 6.
// Round 3's fused attempt (128 freq x 64 frames per block, 256-byte pieces of 128 RGBA rows) ran at 2.7 TB/s: tile rows are
// width * 4 = 2064 / 2080 / 1028 bytes, so no piece starts on a 128-byte line, and the pieces of one row come from blocks far
// apart in time.  Shape by measurement (scripts/ubench/fused_img_shapes.hip: 0.696 ms for the bench's 369 M pixels = 5.3 TB/s,
// against 0.78-0.82 ms for the two kernels; 128 rows per block — one block per CU — 1.0 ms; pieces of 64 frames 1.1-1.2 ms):
//   block = (image, level-0 tile column tx, band of FUSED_FB = 32 image rows): the <= 520 frames of the tile column INCLUDING
//   its 4-frame gutters x 32 bins are staged as u16 in LDS (34 KB: four blocks of 256 threads per CU overlap each other's
//   phases; the real kernel, quantiser included: 32 rows x 256 threads 0.72-0.74 ms, 32 x 512 0.73-0.74, 64 x 1024 0.77-0.79,
//   64 x 512 0.80-0.81, 128 x 1024 0.96-0.97 against 0.81 for the two kernels, same box: profiles/r04_ab_fused_image.txt);
//   read   16-byte loads, 8 lanes per 128-byte row piece, 8 frames per wave-instruction, ALL of a lane's 17 requests in flight
//          before the first is consumed (a block's read phase is one HBM latency, not seventeen);
//   write  the u16 image rows of the column's core frames whole (16 B per lane: 512 frames = one wave-instruction) and every
//          RGBA tile row the band touches — its own tile and, within 4 rows of a tile boundary, the neighbour's gutter — whole,
//          on the destination's 16-byte grid.
// The quantiser and the colour index are the very functions the two kernels use: the results are bit-identical to theirs.
// ------------------------------------------------------------------------------------------
constexpr uint32_t FUSED_PITCH = 536;  // u16 per LDS row: 1072 B = 268 dwords, 268 mod 32 = 12: the eight rows a wave's ds_write_b64 touches land on eight different bank quads
constexpr uint32_t FUSED_COL0 = 8;     // LDS column of the tile column's first CORE frame (16-byte aligned); its left gutter sits in columns 4..7
#if !defined(TH_FUSED_MIN_WAVES)
#define TH_FUSED_MIN_WAVES (FUSED_THREADS / 64)  // four blocks of 256 threads (two of 512) per CU: at most 128 VGPRs
#endif

// The quantiser of the packed path (round 5): the function of quantise<true> in 8 instead of 16 operations per pixel.
//   a = med3(dB - min_dB, a_lo, a_hi)  clamps the dividend into the range whose ends already quantise to 0 and 65535
//       (a_hi = 2 span; a_lo = -(min_value / u16_span) span: u(a_lo) = 0 up to rounding, far below the 0.5 that rounds up),
//       which also takes +-inf and NaN out (v_med3_f32 returns min3 when an operand is NaN: a_lo -> 0, Rust's `NaN as u16`)
//   z = a / span                       correctly rounded (reciprocal + one FMA correction, as quantise<true>)
//   u = z * u16_span + min_value       two roundings, as the reference (contraction is off in this file)
//   r = v_cvt_rpi_i32_f32(u)           floor(u + 0.5) evaluated exactly = f32::round for u >= 0 — checked against
//                                      trunc + (frac >= 0.5) for every f32 in [0, 2^18) on gfx950 (scripts/ubench/rpi_probe.hip:
//                                      0 mismatches, 0.5 - 2^-25 included, where an f32 `u + 0.5` would round up to 1);
//                                      u > -0.5 after the clamp (-> 0); values above 65535 are cut by v_cvt_pk_u16_u32
// Needs u16_span > 0 (two or more colours) and the reciprocal's verified range: the launch-uniform `fastq` says so.
__device__ __forceinline__ uint32_t quantise_regular(float dB, float min_dB, float a_lo, float a_hi, float span, float rinv,
                                                     float u16_span, float min_value) {
    const float a = __builtin_amdgcn_fmed3f(dB - min_dB, a_lo, a_hi);
    const float q0 = a * rinv;
    const float e = __builtin_fmaf(-q0, span, a);
    const float z = __builtin_fmaf(e, rinv, q0);
    const float u = z * u16_span + min_value;
    uint32_t r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(u));
    return r;
}
__device__ __forceinline__ uint32_t pack_u16_sat(uint32_t lo, uint32_t hi) {  // v_cvt_pk_u16_u32: both halves saturate at 65535
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    const us2 p = __builtin_amdgcn_cvt_pk_u16(lo, hi);
    return __builtin_bit_cast(uint32_t, p);
}

// Non-temporal hints (round 5): the kernel's three streams are each touched once — `nt` loads of the spec rows and `nt` stores of
// the image rows and RGBA pieces: 0.732 -> 0.697 ms on a card where the instruction diet above had changed nothing (stores alone
// 0.719; profiles/r05_ab_fused_image_cards.txt).  TH_FUSED_NT: 0 off, 1 stores, 2 stores + loads (default).
#if !defined(TH_FUSED_NT)
#define TH_FUSED_NT 2
#endif
// Raw buffer operations for the three streams of the regular path (round 5): the address is "resource (SGPRs) + scalar offset +
// 32-bit lane offset" — no 64-bit vector address arithmetic — and the cache bits are free to choose: loads `nt` (aux 2), stores
// `nt sc1` (aux 18: write through and do not keep the line).  Card 692537016606: plain 0.735 ms, `nt` global operations 0.650-0.678,
// buffer operations nt / nt 0.652, nt / nt sc1 0.640, nt sc1 / nt sc1 0.636, sc1 alone 0.696 (profiles/r05_ab_fused_image_cards.txt).
#if !defined(TH_FUSED_BUF)
#define TH_FUSED_BUF 1
#endif
#if !defined(TH_FUSED_LD_AUX)
#define TH_FUSED_LD_AUX 2
#endif
#if !defined(TH_FUSED_ST_AUX)
#define TH_FUSED_ST_AUX 18
#endif
typedef uint32_t th_u32x4 __attribute__((ext_vector_type(4)));
typedef float th_f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld_spec4(gptr<const float> p) {  // 16-byte spec load, 16-byte aligned
#if TH_FUSED_NT >= 2
    const th_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<gptr<const th_f32x4>>(p));
    return make_float4(t.x, t.y, t.z, t.w);
#else
    return *reinterpret_cast<gptr<const float4>>(p);
#endif
}
template <bool LUT_IN_LDS>
__global__ __launch_bounds__(FUSED_THREADS, TH_FUSED_MIN_WAVES) void spec_to_img_raster_kernel(
    const FusedJob *__restrict__ jobs, const uint32_t *__restrict__ block_job, uint8_t *const *__restrict__ tiles, float min_dB,
    float span, float u16_span, float min_value, const float *__restrict__ d_range, int all_zero_in,
    const uint32_t *__restrict__ colormap, uint32_t n_colors) {
    constexpr uint32_t FB = FUSED_FB, WAVES = FUSED_THREADS / 64, LPR = FB / 4, GPW = 64 / LPR;
    constexpr uint32_t NIT = 128 / (GPW * WAVES);  // a group = 4 consecutive frames of 4 bins in one lane; 128 groups = the 512 core frames
    static_assert(128 % (GPW * WAVES) == 0 && 512 % FB == 0 && 64 % LPR == 0, "block shape");
    constexpr uint32_t PB = FUSED_PITCH * 2;  // LDS row pitch in bytes
    extern __shared__ __attribute__((aligned(16))) uint16_t ftile[];  // [FB][FUSED_PITCH]
    __shared__ uint32_t lut[LUT_IN_LDS ? 1024 : 1];
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    // (the block's first memory request is its job — two dependent scalar loads; the colour table is requested BEHIND the spec
    // loads and stored to LDS in front of the barrier: filled first, as in round 4, every block began with a memory round trip
    // of its own before it asked for anything else)
    const FusedJob job = jobs[block_job[blockIdx.x]];
    constexpr uint32_t NLUT = LUT_IN_LDS ? 1024 / FUSED_THREADS : 1;
    uint32_t lutv[NLUT];
    auto lut_request = [&]() {
        if constexpr (LUT_IN_LDS) {
#pragma unroll
            for (uint32_t i = 0; i < NLUT; i++) lutv[i] = as_global(colormap)[min(tid + FUSED_THREADS * i, n_colors - 1u)];
        }
    };
    auto lut_store = [&]() {
        if constexpr (LUT_IN_LDS) {
#pragma unroll
            for (uint32_t i = 0; i < NLUT; i++)
                if (tid + FUSED_THREADS * i < n_colors) lut[tid + FUSED_THREADS * i] = lutv[i];
        }
    };
    bool all_zero = all_zero_in != 0;
    float rinv = quantise_rinv(span);
    if (d_range != nullptr) {
        const float lo = d_range[0], hi = d_range[1];
        min_dB = lo;
        span = hi - lo;
        rinv = quantise_rinv(span);
        all_zero = lo == hi && __builtin_isinf(hi) && hi < 0.0f;  // every value -inf: zero image (drawing.rs:16-18)
    }
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 64)  // ablation build: launch + job fetch only
    if (job.n_frames != 0x7fffffffu) return;
#endif
    const gptr<const float> spec = as_global(job.spec);
    const uint32_t out_h = job.i_end - job.i_start, W = job.n_frames;
    const uint32_t local = blockIdx.x - job.first_block;
    const uint32_t band = local % job.n_bands, tx = local / job.n_bands;  // bands of one column are neighbours in the launch
    const uint32_t r0 = band * FB;                                       // first image row of the band (relative to i_start)
    const uint32_t nrows = min(FB, out_h - r0);                          // image rows of this band (0 for an empty image)
    // level-0 tile column tx (render_tiles.rs:290-313): core 512 frames + 4-frame gutters, clipped at the image
    const uint32_t sx = tx * 512u, core = min(W - sx, 512u), ox = sx > 4u ? sx - 4u : 0u, wt = min(W, sx + core + 4u) - ox;
    const uint32_t cs = sx - ox;           // 0 or 4: frames of left gutter
    const uint32_t ct = FUSED_COL0 - cs;   // LDS column of the tile column's first frame `ox`
    // the band lies inside ONE tile row (512 % FB == 0); its rows within 4 of a core boundary are also a neighbour's gutter
    const uint32_t ty0 = r0 / 512u, sy = ty0 * 512u, coreh = min(out_h - sy, 512u);
    const uint32_t oy = sy > 4u ? sy - 4u : 0u, ht = min(out_h, sy + coreh + 4u) - oy;
    uint8_t *const tb0 = (nrows && job.n_ty) ? tiles[job.tile0 + tx * job.n_ty + ty0] : nullptr;  // scalar load, long before its use
    // the band's first four rows are also tile row ty0 - 1's top gutter when the band is the tile row's first, its last four tile
    // row ty0 + 1's bottom gutter when it is the last band of a full tile row: those tiles' pointers, requested here as well
    const bool lo_nb = nrows != 0 && ty0 > 0u && r0 == sy, hi_nb = nrows != 0 && ty0 + 1u < job.n_ty && r0 + FB == sy + 512u;
    uint8_t *const tb_lo = lo_nb ? tiles[job.tile0 + tx * job.n_ty + ty0 - 1u] : nullptr;
    uint8_t *const tb_hi = hi_nb ? tiles[job.tile0 + tx * job.n_ty + ty0 + 1u] : nullptr;

    // ---- read + quantise.  lane -> bins 4 bq .. 4 bq + 3 of the four frames of group g = GPW (wv + WAVES i) + fg
    const uint32_t bq = lane % LPR, fg = lane / LPR;
    const bool al16 = (reinterpret_cast<uintptr_t>(job.spec) & 15u) == 0 && job.spec_pitch % 4u == 0 && (job.i_start + r0) % 4u == 0 &&
                      job.spec_pitch >= 4u;
    const bool fastq = !all_zero && rinv != 0.0f && u16_span > 0.0f;
    // Packed path (block-uniform): 16-byte loads, the reciprocal quantiser, four frames of a bin as one ds_write_b64.
    // REGULAR block: every core frame and every bin of the band exists — no clamps, no selects.  EDGE block (the last tile
    // column, the last band, an image shorter than the band): frames clamped into the image, results outside it zeroed.
    const bool packed = fastq && al16 && job.spec_pitch <= (1u << 20) && nrows > 0;
    const bool regular = core == 512u && nrows == FB && job.i_start + r0 + FB <= job.height;
    auto read_packed = [&](auto edge_tag) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        const float a_hi = span + span, a_lo = -((min_value / u16_span) * span);
        const float us2 = u16_span, mv2 = min_value;
        const uint32_t bin0 = job.i_start + r0 + 4u * bq;
        bool okb[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) okb[k] = !EDGE || (bin0 + k < job.height && 4u * bq + k < nrows);
        // (EDGE: a lane whose bins lie past the row's allocation reads the row's last quad; its bins are all invalid then)
        const uint32_t b16 = EDGE ? min(bin0, job.spec_pitch - 4u) : bin0;
        const uint32_t voff = (4u * fg) * job.spec_pitch + (b16 - (job.i_start + r0));  // elements; 28 * pitch * 4 B < 2^31
        const gptr<const float> b0 = spec + (size_t)sx * job.spec_pitch + (job.i_start + r0);
#if TH_FUSED_BUF
        // (the block's corner of the spec as a buffer resource: 64-bit base in SGPRs, 32-bit offsets — spec_pitch <= 2^20 in the
        // packed path, so 544 rows x 4 MiB stay below 2^32)
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(job.spec + ((size_t)sx * job.spec_pitch + (job.i_start + r0))), 0, -1, 0x00020000);
        (void)b0;
#endif
        float4 v[NIT][4];
#pragma unroll
        for (uint32_t i = 0; i < NIT; i++)
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                const uint32_t f = 4u * GPW * (wv + WAVES * i) + q;  // frame - sx - 4 fg (wave-uniform)
                if constexpr (!EDGE) {
                    const gptr<const float> rowp = b0 + (size_t)f * job.spec_pitch;  // scalar
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 1)  // ablation build: no spec reads
                    v[i][q] = make_float4(-1.0f * (float)lane, -2.0f * f, -3.0f, -4.0f * fg);
                    if (min_dB == 12345.0f)
#endif
#if TH_FUSED_BUF  // raw buffer load: scalar row offset + per-lane offset, `nt`
                    {
                        const th_u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs_spec, voff * 4u, f * (job.spec_pitch * 4u), TH_FUSED_LD_AUX);
                        // (scalar copies first: __builtin_bit_cast applied to a vector-element lvalue reads element 0 for every element)
                        const uint32_t t0 = t.x, t1 = t.y, t2 = t.z, t3 = t.w;
                        v[i][q] = make_float4(__builtin_bit_cast(float, t0), __builtin_bit_cast(float, t1), __builtin_bit_cast(float, t2), __builtin_bit_cast(float, t3));
                    }
#else
                    v[i][q] = ld_spec4(rowp + voff);
#endif
                } else {
                    const uint32_t fc = min(sx + f + 4u * fg, W - 1u);
                    v[i][q] = ld_spec4(spec + ((size_t)fc * job.spec_pitch + b16));
                }
            }
        // the eight gutter frames x FB bins: one 4-byte load per thread (a gutter the tile column does not have is never read
        // back: its address is clamped into the image and whatever it quantises to stays in LDS unused)
        constexpr uint32_t NGV = (8 * FB + FUSED_THREADS - 1) / FUSED_THREADS;
        float gv[NGV];
#pragma unroll
        for (uint32_t e = 0; e < NGV; e++) {
            const uint32_t id = tid + FUSED_THREADS * e, g = id / FB, bin = id % FB;
            const uint32_t fr = g < 4u ? (sx >= 4u ? sx - 4u + g : 0u) : min(sx + 508u + g, W - 1u);
            const uint32_t bn = EDGE ? min(job.i_start + r0 + bin, job.height - 1u) : job.i_start + r0 + bin;
            gv[e] = (8 * FB % FUSED_THREADS == 0 || id < 8 * FB) ? spec[(size_t)fr * job.spec_pitch + bn] : 0.0f;
        }
        lut_request();
        const uint32_t wbase = (4u * bq) * PB + 2u * (FUSED_COL0 + 4u * (GPW * wv + fg));  // LDS byte address of (row 4 bq, group of i = 0)
#pragma unroll
        for (uint32_t i = 0; i < NIT; i++) {
            uint32_t u[4][4];
#pragma unroll
            for (uint32_t q = 0; q < 4; q++) {
                u[q][0] = quantise_regular(v[i][q].x, min_dB, a_lo, a_hi, span, rinv, us2, mv2);
                u[q][1] = quantise_regular(v[i][q].y, min_dB, a_lo, a_hi, span, rinv, us2, mv2);
                u[q][2] = quantise_regular(v[i][q].z, min_dB, a_lo, a_hi, span, rinv, us2, mv2);
                u[q][3] = quantise_regular(v[i][q].w, min_dB, a_lo, a_hi, span, rinv, us2, mv2);
                if constexpr (EDGE) {  // frames past the image and bins past the spec are zero (drawing.rs:24-31, the row padding)
                    const bool okf = sx + 4u * GPW * (wv + WAVES * i) + q + 4u * fg < W;
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) u[q][k] = (okf && okb[k]) ? u[q][k] : 0u;
                }
            }
#pragma unroll
            for (uint32_t k = 0; k < 4; k++)
                *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(ftile) + wbase + k * PB + 8u * GPW * WAVES * i) =
                    make_uint2(pack_u16_sat(u[0][k], u[1][k]), pack_u16_sat(u[2][k], u[3][k]));
        }
#pragma unroll
        for (uint32_t e = 0; e < NGV; e++) {
            const uint32_t id = tid + FUSED_THREADS * e, g = id / FB, bin = id % FB;
            if (8 * FB % FUSED_THREADS == 0 || id < 8 * FB) {
                uint32_t uq = min(quantise_regular(gv[e], min_dB, a_lo, a_hi, span, rinv, us2, mv2), 65535u);
                if (EDGE && job.i_start + r0 + bin >= job.height) uq = 0u;
                ftile[bin * FUSED_PITCH + (g < 4u ? 4u + g : FUSED_COL0 + 508u + g)] = (uint16_t)uq;
            }
        }
        lut_store();
    };
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 8)  // ablation build: no read / quantise phase
    if (min_dB == 12345.0f)
#endif
    if (packed && regular) {
        read_packed(std::false_type{});
    } else if (packed) {
        read_packed(std::true_type{});
    } else {
        // ---- anything else (odd pitches / bases, slow division, one colour, the zero image): element-wise — frame
        // ox + tt -> LDS column ct + tt, lane -> bins fl .. fl + 3 of frame ox + FPI (wv + WAVES i) + fr, element-wise ds_write_b16
        constexpr uint32_t FPI = GPW, NLD = (FUSED_PITCH + FPI * WAVES - 1) / (FPI * WAVES);
        const uint32_t fl = 4u * bq, fr = fg;
        const uint32_t bin0 = job.i_start + r0 + fl;
        bool okb[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) okb[k] = bin0 + k < job.height && r0 + fl + k < out_h;
        // 16-byte loads: aligned base / pitch / first bin (block-uniform).  A lane whose four bins lie past the row's allocation
        // reads the row's last quad instead — with these alignments its bins are all >= spec_pitch >= height, so they are discarded
        // anyway — which keeps the loads free of per-lane branches (a branch around a load makes the compiler wait for the previous one)
        const uint32_t b16 = min(bin0, job.spec_pitch - 4u);
        const float qnan = __builtin_nanf("");
        lut_request();
        lut_store();
        // (in rounds of four loads: this path is rare, keep its registers below the regular path's)
#pragma unroll 1
        for (uint32_t i0 = 0; i0 < NLD; i0 += 4) {
            float v[4][4];
#pragma unroll
            for (uint32_t ii = 0; ii < 4; ii++) {
                const uint32_t tt = FPI * (wv + WAVES * (i0 + ii)) + fr;
                const gptr<const float> rowp = spec + (size_t)min(ox + tt, W - 1u) * job.spec_pitch;
                if (al16) {
                    const float4 x = *reinterpret_cast<gptr<const float4>>(rowp + b16);
                    v[ii][0] = x.x;
                    v[ii][1] = x.y;
                    v[ii][2] = x.z;
                    v[ii][3] = x.w;
                } else {
#pragma unroll
                    for (uint32_t k = 0; k < 4; k++) v[ii][k] = rowp[min(bin0 + k, max(job.height, 1u) - 1u)];
                }
            }
            // the three cases are launch-uniform: one specialised loop each instead of a branch per element (as spec_to_img_kernel)
            // (NaN -> 0: rows >= H and columns past the image are zero; columns >= wt are the row padding the last tile column writes)
#define TH_FUSED_QUANT(EXPR)                                                                             \
    _Pragma("unroll") for (uint32_t ii = 0; ii < 4; ii++) {                                              \
        const uint32_t tt = FPI * (wv + WAVES * (i0 + ii)) + fr;                                         \
        if (ct + tt < FUSED_PITCH) {                                                                     \
            const bool okt = tt < wt;                                                                    \
            _Pragma("unroll") for (uint32_t k = 0; k < 4; k++) {                                         \
                const float x = (okt && okb[k]) ? v[ii][k] : qnan;                                       \
                (void)x;                                                                                 \
                ftile[(fl + k) * FUSED_PITCH + ct + tt] = (uint16_t)(EXPR);                              \
            }                                                                                            \
        }                                                                                                \
    }
            if (all_zero) {
                TH_FUSED_QUANT(0u)
            } else if (rinv != 0.0f) {
                TH_FUSED_QUANT(quantise<true>(x, min_dB, span, rinv, u16_span, min_value))
            } else {
                TH_FUSED_QUANT(quantise<false>(x, min_dB, span, rinv, u16_span, min_value))
            }
#undef TH_FUSED_QUANT
        }
    }
    __syncthreads();
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 128)  // ablation build: stop behind the barrier
    if (job.n_frames != 0x7fffffffu) return;
#endif
    // ---- u16 image rows: the column's core frames [sx, sx + core) (+ the row padding in the last column), 8 px per lane
    const gptr<uint16_t> img = as_global(job.img);
    // rows at the library's padded pitch own their padding (see spec_to_img_kernel): complete the last 128-byte line
    const uint32_t t_lim = (job.img_pitch % IMG_TILE_T == 0 && job.img_pitch - W < IMG_TILE_T) ? job.img_pitch : W;
    const uint32_t c_lim = min(t_lim - sx, 512u);  // columns of this tile column to write (the last column: up to the pitch)
    const bool img_al = (reinterpret_cast<uintptr_t>(job.img) & 15u) == 0 && job.img_pitch % 8u == 0;
#if TH_FUSED_BUF
    // (the band's corner of the image as a buffer resource; 32 rows x img_pitch x 2 bytes below 2^32)
    const bool img_buf = job.img_pitch <= (1u << 24);
    const __amdgpu_buffer_rsrc_t rs_img = __builtin_amdgcn_make_buffer_rsrc(job.img + ((size_t)r0 * job.img_pitch + sx), 0, -1, 0x00020000);
#endif
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 32)  // ablation build: no image-row phase
    if (min_dB == 12345.0f)
#endif
#pragma unroll
    for (uint32_t i = 0; i < (FB + WAVES - 1) / WAVES; i++) {
        const uint32_t r = wv + WAVES * i, c = 8u * lane;
        if (r < nrows && c < c_lim) {
            const uint16_t *src = &ftile[r * FUSED_PITCH + FUSED_COL0 + c];  // 16-byte aligned
            const gptr<uint16_t> dst = img + (size_t)(r0 + r) * job.img_pitch + sx + c;
            if (img_al && c + 8u <= c_lim) {
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 2)  // ablation build: no u16 image stores
                if (reinterpret_cast<const uint4 *>(src)->x == 0x12345678u)
#endif
#if TH_FUSED_BUF
                if (img_buf) __builtin_amdgcn_raw_buffer_store_b128(*reinterpret_cast<const th_u32x4 *>(src), rs_img, (r * job.img_pitch + c) * 2u, 0, TH_FUSED_ST_AUX);
                else
                    __builtin_nontemporal_store(*reinterpret_cast<const th_u32x4 *>(src), reinterpret_cast<gptr<th_u32x4>>(dst));
#elif TH_FUSED_NT >= 1
                __builtin_nontemporal_store(*reinterpret_cast<const th_u32x4 *>(src), reinterpret_cast<gptr<th_u32x4>>(dst));
#else
                *reinterpret_cast<gptr<uint4>>(dst) = *reinterpret_cast<const uint4 *>(src);
#endif
            } else {
                for (uint32_t k = 0; k < 8u && c + k < c_lim; k++) dst[k] = src[k];
            }
        }
    }
    // ---- RGBA rows: image row r_img -> row (oy + h - 1 - r_img) of every tile (tx, ty) that holds it
    auto look = [&](uint32_t val) -> uint32_t {
        const uint32_t ci = colour_index(val, n_colors);
        if constexpr (LUT_IN_LDS) return lut[ci];
        else return as_global(colormap)[ci];
    };
    // The band's rows in its OWN tile are one contiguous piece of that tile (rows oy + ht - 1 - r_img, descending with r_img):
    // when the rows are whole quads (wt % 4 == 0, 16-byte aligned tile) the piece is written front to back, 16 bytes per lane,
    // a wave-instruction = 1 KiB of consecutive addresses whatever the row length (round 4 went row by row: three
    // wave-instructions per 2080-byte row, the third with two lanes)
    // (wt >= 32: at least eight quads per row, so that a lane's "quad before the piece" lies in the one row before it)
    const bool flat = tb0 != nullptr && wt % 4u == 0 && wt >= 32u && (reinterpret_cast<uintptr_t>(tb0) & 15u) == 0 && nrows > 0;
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 16)  // ablation build: no RGBA phase
    if (min_dB == 12345.0f)
#endif
    if (flat) {
        const uint32_t qpr = wt >> 2, nq = nrows * qpr;
        const uint32_t r_top = oy + ht - 1u - (r0 + nrows - 1u);  // tile row of the band's LAST image row = the piece's first row
        const gptr<uint4> dst = reinterpret_cast<gptr<uint4>>(as_global(reinterpret_cast<uint32_t *>(tb0)) + (size_t)r_top * wt);
#if TH_FUSED_BUF
        const __amdgpu_buffer_rsrc_t rs_rgba = __builtin_amdgcn_make_buffer_rsrc(tb0 + (size_t)r_top * wt * 4u, 0, -1, 0x00020000);
        (void)dst;
#endif
        const uint32_t dj = FUSED_THREADS / qpr, dc = FUSED_THREADS % qpr;  // block-uniform
        // The piece starts wherever its first row starts: at a multiple of 16 bytes, k quads into a 128-byte line.  Quads are
        // dealt to the lanes in LINE-ALIGNED order — lane t of round m takes quad qa - k of the piece, qa = t + THREADS m —
        // so that a wave-instruction writes eight whole lines (round 5: dealt from the piece's first quad, every 1 KiB store
        // straddled nine lines and every line boundary was written by two different waves)
        const uint32_t k = (uint32_t)(reinterpret_cast<uintptr_t>(tb0) / 16u + (size_t)r_top * qpr) & 7u;
        const int32_t q0 = (int32_t)tid - (int32_t)k;                       // first quad of this lane (negative: none in round 0)
        uint32_t j = q0 >= 0 ? (uint32_t)q0 / qpr : 0xffffffffu;            // piece row; -1 = "the row before the piece" (mod 2^32)
        uint32_t c = q0 >= 0 ? (uint32_t)q0 - j * qpr : (uint32_t)(q0 + (int32_t)qpr);  // quad of the row
        // LDS byte address of the quad: image row (nrows - 1 - j), column ct + 4 c
        uint32_t la = (nrows - 1u - j) * PB + 2u * ct + 8u * c;
        const uint32_t dla = 8u * dc - dj * PB;  // (mod 2^32)
        // rounds of four quads per lane: the four u16 quads are requested together, then the sixteen LUT entries, then the four
        // 16-byte stores — two LDS latencies per round instead of eight
        constexpr uint32_t NQI = (FB * 130u + 7u + FUSED_THREADS - 1u) / FUSED_THREADS;
#pragma unroll 1
        for (uint32_t m0 = 0; m0 < NQI; m0 += 4) {
            if (m0 * FUSED_THREADS >= nq + k) break;  // block-uniform
            uint2 w[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t q = tid + (m0 + u) * FUSED_THREADS - k;  // (mod 2^32: a negative quad is >= nq)
                w[u] = *reinterpret_cast<const uint2 *>(reinterpret_cast<const char *>(ftile) + (q < nq ? la : 0u));
                c += dc;
                la += dla;
                if (c >= qpr) {
                    c -= qpr;
                    la -= 8u * qpr + PB;
                }
            }
            uint4 o[4];
#pragma unroll
            for (uint32_t u = 0; u < 4; u++)
                o[u] = make_uint4(look(w[u].x & 0xffffu), look(w[u].x >> 16), look(w[u].y & 0xffffu), look(w[u].y >> 16));
#pragma unroll
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t q = tid + (m0 + u) * FUSED_THREADS - k;
#if defined(TH_FUSED_ABL) && (TH_FUSED_ABL & 4)  // ablation build: no RGBA stores
                if (o[u].x == 0x12345678u && o[u].y == 0x9abcdef0u)
#endif
#if TH_FUSED_BUF
                if (q < nq) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(th_u32x4, o[u]), rs_rgba, q * 16u, 0, TH_FUSED_ST_AUX);
#elif TH_FUSED_NT >= 1
                if (q < nq) __builtin_nontemporal_store(__builtin_bit_cast(th_u32x4, o[u]), reinterpret_cast<gptr<th_u32x4>>(dst) + q);
#else
                if (q < nq) dst[q] = o[u];
#endif
            }
        }
    }
    // rows that are ALSO a neighbour's gutter (the first / last 4 rows of a tile row's core), and the own tile where the flat
    // form does not apply (odd widths: the last tile column): row by row.  The neighbour tiles' pointers were requested at the
    // top of the block: inside the row loop each was a dependent scalar load — a memory round trip per (row, neighbour) probe
    // in the one block in eight that sits at a tile boundary, 0.08 ms of the launch's 0.28 ms without memory traffic.
    auto row_to_tile = [&](uint32_t r, uint8_t *tbase, uint32_t ty) __attribute__((always_inline)) {
        if (tbase == nullptr) return;
        const uint32_t r_img = r0 + r;
        const uint32_t sy2 = ty * 512u, coreh2 = min(out_h - sy2, 512u);
        const uint32_t oy2 = sy2 > 4u ? sy2 - 4u : 0u, ht2 = min(out_h, sy2 + coreh2 + 4u) - oy2;
        if (r_img < oy2 || r_img >= oy2 + ht2) return;
        const gptr<uint32_t> dst = as_global(reinterpret_cast<uint32_t *>(tbase)) + (size_t)(oy2 + ht2 - 1u - r_img) * wt;
        const uint16_t *const srow = &ftile[r * FUSED_PITCH + ct];
        // 16-byte stores on the destination's 16-byte grid (rows of odd widths start 4 / 8 / 12 bytes off it)
        const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(tbase) / 4u + (size_t)(oy2 + ht2 - 1u - r_img) * wt) & 3u;
#pragma unroll
        for (uint32_t k = 0; k < 3; k++) {  // wt <= 520: at most 131 quads (+ 1 for a shifted grid)
            const int32_t c = (int32_t)(4u * (lane + 64u * k)) - (int32_t)mis;
            if (c >= (int32_t)wt) continue;
            if (c >= 0 && (uint32_t)c + 4u <= wt) {
                uint32_t p0, p1, p2, p3;
                if (mis == 0u) {  // (wave-uniform) the source quad is 8-byte aligned: one LDS read
                    const uint2 q = *reinterpret_cast<const uint2 *>(srow + c);
                    p0 = q.x & 0xffffu;
                    p1 = q.x >> 16;
                    p2 = q.y & 0xffffu;
                    p3 = q.y >> 16;
                } else {
                    p0 = srow[c];
                    p1 = srow[c + 1];
                    p2 = srow[c + 2];
                    p3 = srow[c + 3];
                }
                *reinterpret_cast<gptr<uint4>>(dst + c) = make_uint4(look(p0), look(p1), look(p2), look(p3));
            } else {
                for (int32_t q = c < 0 ? 0 : c; q < c + 4 && q < (int32_t)wt; q++) dst[q] = look(srow[q]);
            }
        }
    };
    if (nrows == 0) return;
    if (!flat) {  // the band's own tile, row by row
#pragma unroll 1
        for (uint32_t r = wv; r < nrows; r += WAVES) row_to_tile(r, tb0, ty0);
    }
    if (lo_nb) {  // image rows sy .. sy + 3 = the band's rows 0 .. 3 (the band is the tile row's first): tile (tx, ty0 - 1)'s top gutter
#pragma unroll 1
        for (uint32_t r = wv; r < min(4u, nrows); r += WAVES) row_to_tile(r, tb_lo, ty0 - 1u);
    }
    if (hi_nb) {  // image rows sy + 508 .. sy + 511 = the band's last four rows (the band is a full tile row's last): tile (tx, ty0 + 1)'s bottom gutter
#pragma unroll 1
        for (uint32_t r = FB - 4u + wv; r < min(FB, nrows); r += WAVES) row_to_tile(r, tb_hi, ty0 + 1u);
    }
}

hipError_t launch_spec_to_img_raster(const FusedJob *d_jobs, const uint32_t *d_block_job, uint32_t n_blocks, uint8_t *const *d_tiles,
                                     float min_dB, float max_dB, const float *d_range, int all_zero, const uint8_t *d_colormap,
                                     uint32_t n_colors, hipStream_t s) {
    if (!n_blocks) return hipSuccess;
    uint32_t min_value = 1;  // drawing.rs:20-22, as launch_spec_to_img
    if (n_colors) {
        const double r = __builtin_round(65535.0 / (double)n_colors);
        const uint32_t v = r >= 65535.0 ? 65535u : (uint32_t)r;
        min_value = v > 1 ? v : 1;
    }
    const float span = max_dB - min_dB;
    const size_t lds = (size_t)FUSED_FB * FUSED_PITCH * sizeof(uint16_t);
    const bool lut_lds = n_colors <= 1024;
    auto kern = lut_lds ? spec_to_img_raster_kernel<true> : spec_to_img_raster_kernel<false>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(n_blocks), dim3(FUSED_THREADS), lds, s, d_jobs, d_block_job, d_tiles, min_dB, span,
                       (float)(65535u - min_value), (float)min_value, d_range, all_zero,
                       reinterpret_cast<const uint32_t *>(d_colormap), n_colors);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// LOD > 0 tiles: separable Lanczos3 in f64 with host-tabulated taps (see LodAxis).  The reference
// delegates this to fast_image_resize 6.0.0, whose source is not vendored: parity is pinned only at
// level (0,0) (exact copy) and by three coarse tests; this is the textbook filter, bit-identical to
// the CPU restatement in oracle/ (same taps, same summation order, one rounding per pass).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ uint16_t lod_round(double acc, double wsum) {
    double v = wsum != 0.0 ? acc / wsum : 0.0;
    v = floor(v + 0.5);
    if (v < 0.0) v = 0.0;
    if (v > 65535.0) v = 65535.0;
    return (uint16_t)v;
}

__global__ __launch_bounds__(256) void lod_hpass_kernel(const uint16_t *__restrict__ img, uint32_t img_pitch,
                                                        uint32_t y_lo, LodAxis ax, uint16_t *__restrict__ tmp,
                                                        uint32_t tmp_pitch) {
    const uint32_t ox = blockIdx.x * 256 + threadIdx.x, r = blockIdx.y;
    if (ox >= ax.n_out) return;
    const gptr<const uint16_t> row = as_global(img) + (size_t)(y_lo + r) * img_pitch + ax.start[ox];
    const gptr<const double> w = as_global(ax.w) + (size_t)ox * ax.max_taps;
    const int32_t n = ax.count[ox];
    double acc = 0.0;
    for (int32_t t = 0; t < n; t++) acc += w[t] * (double)row[t];
    as_global(tmp)[(size_t)r * tmp_pitch + ox] = lod_round(acc, ax.wsum[ox]);
}

__global__ __launch_bounds__(256) void lod_vpass_kernel(const uint16_t *__restrict__ tmp, uint32_t tmp_pitch, uint32_t y_lo,
                                                        LodAxis ay, uint32_t dw, uint16_t *__restrict__ lod,
                                                        uint32_t lod_pitch) {
    const uint32_t ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= dw) return;
    const gptr<const uint16_t> col = as_global(tmp) + (size_t)(ay.start[oy] - (int32_t)y_lo) * tmp_pitch + ox;
    const gptr<const double> w = as_global(ay.w) + (size_t)oy * ay.max_taps;
    const int32_t n = ay.count[oy];
    double acc = 0.0;
    for (int32_t t = 0; t < n; t++) acc += w[t] * (double)col[(size_t)t * tmp_pitch];
    as_global(lod)[(size_t)oy * lod_pitch + ox] = lod_round(acc, ay.wsum[oy]);
}

// The vertical pass over MANY images of one shape in one launch (the mip pyramids of a batch of tracks: a launch per
// level instead of a launch per level and channel; the horizontal pass is this one on the transposed image, see
// transpose_u16_batch_kernel); blockIdx.z picks the job.
// A thread owns one column and R consecutive output rows.  Their tap windows overlap (a level-l output spans 6 * 2^l
// source rows, its neighbour starts 2^l further down), so the thread walks the union of the windows once — one u16 load
// and one conversion per source row, R independent accumulators — where a thread per output loaded and converted every
// row six times and waited on one dependent f64 chain (5.8 ms per pyramid rebuild of 32 images, 92 % of set_dB_range).
// The block's taps sit in LDS as R rows over the union, ZERO outside each output's own window: the inner loop is then
// branch-free (acc + 0.0 * v == acc exactly: v is a finite non-negative pixel), reads its taps as broadcasts, and every
// output still adds its taps in ascending order, mul then add: bit-identical to lod_vpass_kernel and to the oracle.
// R = 8 / 4 / 2 by what fits 64 KB of LDS and leaves the launch a thousand blocks; else one output per thread, taps from
// the global table (lod_vpass_batch1_kernel).
template <uint32_t R>
__global__ __launch_bounds__(256) void lod_vpass_batch_kernel(const LodPassJob *__restrict__ jobs, LodAxis ay, uint32_t dw, uint32_t span) {
    extern __shared__ __attribute__((aligned(16))) double lod_wz[];  // [R][span]
    const uint32_t oy0 = blockIdx.y * R;
    const uint32_t n_rows = min(R, ay.n_out - oy0);
    int32_t st[R], cn[R];  // block-uniform
    int32_t t_hi = 0;
#pragma unroll
    for (uint32_t o = 0; o < R; o++) {
        const uint32_t oy = min(oy0 + o, ay.n_out - 1u);
        st[o] = __builtin_amdgcn_readfirstlane(ay.start[oy]);
        cn[o] = o < n_rows ? __builtin_amdgcn_readfirstlane(ay.count[oy]) : 0;
        t_hi = max(t_hi, st[o] + cn[o]);
    }
    const int32_t t_lo = st[0];  // (starts ascend: the first row any of the outputs needs)
    const uint32_t len = min((uint32_t)(t_hi - t_lo), span);  // == t_hi - t_lo (span is the maximum over the axis)
#pragma unroll
    for (uint32_t o = 0; o < R; o++) {
        const gptr<const double> wo = as_global(ay.w) + (size_t)min(oy0 + o, ay.n_out - 1u) * ay.max_taps;
        for (uint32_t j = threadIdx.x; j < len; j += 256) {
            const int32_t rel = t_lo + (int32_t)j - st[o];
            lod_wz[o * span + j] = (uint32_t)rel < (uint32_t)cn[o] ? wo[rel] : 0.0;
        }
    }
    __syncthreads();
    const uint32_t ox = blockIdx.x * 256 + threadIdx.x;
    if (ox >= dw) return;
    const LodPassJob job = jobs[blockIdx.z];
    double acc[R];
#pragma unroll
    for (uint32_t o = 0; o < R; o++) acc[o] = 0.0;
    const gptr<const uint16_t> col = as_global(job.src) + (size_t)t_lo * job.src_pitch + ox;
#pragma unroll 4
    for (uint32_t j = 0; j < len; j++) {
        const double v = (double)col[(size_t)j * job.src_pitch];
#pragma unroll
        for (uint32_t o = 0; o < R; o++) acc[o] += lod_wz[o * span + j] * v;
    }
#pragma unroll
    for (uint32_t o = 0; o < R; o++)
        if (o < n_rows) as_global(job.dst)[(size_t)(oy0 + o) * job.dst_pitch + ox] = lod_round(acc[o], ay.wsum[oy0 + o]);
}
__global__ __launch_bounds__(256) void lod_vpass_batch1_kernel(const LodPassJob *__restrict__ jobs, LodAxis ay, uint32_t dw) {
    const uint32_t ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= dw) return;
    const LodPassJob job = jobs[blockIdx.z];
    const gptr<const uint16_t> col = as_global(job.src) + (size_t)ay.start[oy] * job.src_pitch + ox;
    const gptr<const double> w = as_global(ay.w) + (size_t)oy * ay.max_taps;
    const int32_t n = ay.count[oy];
    double acc = 0.0;
    for (int32_t t = 0; t < n; t++) acc += w[t] * (double)col[(size_t)t * job.src_pitch];
    as_global(job.dst)[(size_t)oy * job.dst_pitch + ox] = lod_round(acc, ay.wsum[oy]);
}
// u16 transpose of many images of one shape: dst[x][y] = src[y][x], 64 x 64 tiles through LDS (both sides coalesced).
// The mip builder runs the horizontal Lanczos pass as a vertical pass over the transposed image: a thread per output
// column that walks its taps along a row reads a different cache line in every lane (the upper levels, hundreds of taps
// per output, took 1-3 ms each for 32 images), a thread that walks them down a column reads 128 contiguous bytes per wave.
__global__ __launch_bounds__(256) void transpose_u16_batch_kernel(const LodPassJob *__restrict__ jobs, uint32_t w, uint32_t h) {
    __shared__ uint16_t tile[64][66];
    const LodPassJob job = jobs[blockIdx.z];
    const uint32_t x0 = blockIdx.x * 64, y0 = blockIdx.y * 64;
    const uint32_t tx = threadIdx.x & 63u, ty = threadIdx.x >> 6;
    for (uint32_t j = ty; j < 64; j += 4) {
        const uint32_t x = x0 + tx, y = y0 + j;
        tile[j][tx] = (x < w && y < h) ? as_global(job.src)[(size_t)y * job.src_pitch + x] : (uint16_t)0;
    }
    __syncthreads();
    for (uint32_t j = ty; j < 64; j += 4) {
        const uint32_t x = x0 + j, y = y0 + tx;
        if (x < w && y < h) as_global(job.dst)[(size_t)x * job.dst_pitch + y] = tile[tx][j];
    }
}
hipError_t launch_transpose_u16_batch(const LodPassJob *d_jobs, uint32_t n_jobs, uint32_t w, uint32_t h, hipStream_t s) {
    if (!n_jobs || !w || !h) return hipSuccess;
    if (n_jobs > 65535u || (h + 63) / 64 > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(transpose_u16_batch_kernel, dim3((w + 63) / 64, (h + 63) / 64, n_jobs), dim3(256), 0, s, d_jobs, w, h);
    return hipGetLastError();
}

hipError_t launch_lod_vpass_batch(const LodPassJob *d_jobs, uint32_t n_jobs, LodAxis ay, uint32_t dw, hipStream_t s) {
    if (!n_jobs || !ay.n_out || !dw) return hipSuccess;
    if (n_jobs > 65535u) return hipErrorInvalidValue;
    const uint32_t bx = (dw + 255) / 256;
    uint32_t r = 1, span = 0;
    for (int k = 0; k < 3; k++) {  // R = 8, 4, 2: the largest that fits the LDS and keeps ~1000 blocks in the launch
        const uint32_t rk = 8u >> k;
        const uint64_t blocks = (uint64_t)bx * ((ay.n_out + rk - 1) / rk) * n_jobs;
        if (ay.span[k] != 0 && (size_t)rk * ay.span[k] * sizeof(double) <= 64 * 1024 && (blocks >= 1024 || rk == 2)) {
            r = rk;
            span = ay.span[k];
            break;
        }
    }
    const uint32_t n_blk = (ay.n_out + r - 1) / r;
    if (n_blk > 65535u) return hipErrorInvalidValue;
    const dim3 grid(bx, n_blk, n_jobs);
    const size_t lds = (size_t)r * span * sizeof(double);
    if (r == 8) hipLaunchKernelGGL((lod_vpass_batch_kernel<8>), grid, dim3(256), lds, s, d_jobs, ay, dw, span);
    else if (r == 4) hipLaunchKernelGGL((lod_vpass_batch_kernel<4>), grid, dim3(256), lds, s, d_jobs, ay, dw, span);
    else if (r == 2) hipLaunchKernelGGL((lod_vpass_batch_kernel<2>), grid, dim3(256), lds, s, d_jobs, ay, dw, span);
    else hipLaunchKernelGGL(lod_vpass_batch1_kernel, grid, dim3(256), 0, s, d_jobs, ay, dw);
    return hipGetLastError();
}

hipError_t launch_lod_hpass(const uint16_t *d_img, uint32_t img_pitch, uint32_t y_lo, uint32_t n_rows, LodAxis ax,
                            uint16_t *d_tmp, uint32_t tmp_pitch, hipStream_t s) {
    if (!n_rows || !ax.n_out) return hipSuccess;
    hipLaunchKernelGGL(lod_hpass_kernel, dim3((ax.n_out + 255) / 256, n_rows), dim3(256), 0, s, d_img, img_pitch, y_lo, ax,
                       d_tmp, tmp_pitch);
    return hipGetLastError();
}

hipError_t launch_lod_vpass(const uint16_t *d_tmp, uint32_t tmp_pitch, uint32_t y_lo, LodAxis ay, uint32_t dw,
                            uint16_t *d_lod, uint32_t lod_pitch, hipStream_t s) {
    if (!ay.n_out || !dw) return hipSuccess;
    hipLaunchKernelGGL(lod_vpass_kernel, dim3((dw + 255) / 256, ay.n_out), dim3(256), 0, s, d_tmp, tmp_pitch, y_lo, ay, dw,
                       d_lod, lod_pitch);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Streaming device-to-device copy, one 16-byte load and store per thread, a block per 4 KiB: the yardstick bench.py
// measures beside every roofline figure (what a pure read + write stream reaches on this box, this run; the shape
// scripts/ubench/copy_rate.hip found fastest: 6.2 TB/s, a persistent grid-stride loop 4.9).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void copy_f4_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) as_global(dst)[i] = as_global(src)[i];
}
hipError_t launch_copy_f4(const void *d_src, void *d_dst, uint64_t bytes, hipStream_t s) {
    const uint64_t n = bytes / 16;
    if (!n) return hipSuccess;
    if ((n + 255) / 256 > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_f4_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s, (const uint4 *)d_src, (uint4 *)d_dst, n);
    return hipGetLastError();
}

}  // namespace th
