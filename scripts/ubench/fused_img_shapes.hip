// Micro-benchmark (development tool, round 4): access structures for a FUSED quantise + level-0 raster pass
//   f32 spec [T][HP] (frame-major)  ->  u16 image [H][TP] (transposed)  +  RGBA level-0 tiles (512 px core + 4 px gutters,
//   rows flipped, one contiguous w * h * 4 byte array per tile, render_tiles.rs:290-351)
// without the arithmetic: 10 algorithmic bytes per pixel instead of the 6 + 6 of the two kernels.  Which block shape
// streams?  (Round 3's fused attempt wrote 256-byte pieces of 128 RGBA rows per block and ran at 2.7 TB/s.)
//   mode 0: the two-kernel traffic as it is today (quantise shape 128 freq x 64 frames, then a linear raster) — reference
//   mode 1: block = 128 freq x 64 frames, RGBA written as 256-byte row pieces straight from that block (round 3's variant)
//   mode 2: block = FB freq x one whole tile column (<= 520 frames) staged in LDS, u16 rows and RGBA rows written whole
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

constexpr int T = 2813, H = 1025, HP = 1056, TP = 2816, N = 128;
constexpr int NTX = (T + 511) / 512, NTY = (H + 511) / 512;

struct Geom {  // level-0 tile geometry of one image
    int ox[NTX], w[NTX], oy[NTY], h[NTY];
    size_t base[NTX][NTY];  // byte offset of tile (tx, ty) inside the image's RGBA block (256-byte aligned)
    size_t total;
};
// aligned: an artificial geometry without gutters and with tile widths that are multiples of 64 px — every 64-frame piece of
// an RGBA row is then two whole 128-byte lines: tells whether round 3's fused variant died of its misaligned pieces
static Geom make_geom(bool aligned = false) {
    Geom g{};
    for (int tx = 0; tx < NTX; tx++) {
        const int s = tx * 512, c = (T - s < 512 ? T - s : 512);
        g.ox[tx] = aligned ? s : (s > 4 ? s - 4 : 0);
        g.w[tx] = aligned ? (c + 63) / 64 * 64 : (s + c + 4 < T ? s + c + 4 : T) - g.ox[tx];
    }
    for (int ty = 0; ty < NTY; ty++) {
        const int s = ty * 512, c = (H - s < 512 ? H - s : 512);
        g.oy[ty] = aligned ? s : (s > 4 ? s - 4 : 0);
        g.h[ty] = aligned ? c : (s + c + 4 < H ? s + c + 4 : H) - g.oy[ty];
    }
    size_t off = 0;
    for (int tx = 0; tx < NTX; tx++)
        for (int ty = 0; ty < NTY; ty++) {
            g.base[tx][ty] = off;
            off += ((size_t)g.w[tx] * g.h[ty] * 4 + 255) / 256 * 256;
        }
    g.total = off;
    return g;
}
__constant__ Geom G;

__device__ __forceinline__ uint32_t lutc(const uint32_t *lut, uint32_t v) { return lut[v >> 8]; }

// ---- mode 2: FB freq rows x one tile column (frames [ox, ox + w)) per block, staged in LDS
template <int FB>
__global__ __launch_bounds__(1024) void fused_cols(const float *__restrict__ spec, uint16_t *__restrict__ img, uint8_t *__restrict__ rgba,
                                                   const uint32_t *__restrict__ cmap) {
    constexpr int PITCH = 528 + 2;             // u16 per LDS row: 520 frames + pad (odd dword count)
    extern __shared__ uint16_t tile[];         // [FB][PITCH]
    __shared__ uint32_t lut[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;  // 16 waves
    if (tid < 256) lut[tid] = cmap[tid];
    constexpr int BANDS = (H + FB - 1) / FB;
    const int b = blockIdx.x;
    const int n = b / (BANDS * NTX);
    const int l = b % (BANDS * NTX);
    const int band = l % BANDS, tx = l / BANDS;   // frequency-fastest
    const int f0 = band * FB, x0 = G.ox[tx], w = G.w[tx];
    const float *sp = spec + (size_t)n * T * HP;
    // read: a wave-instruction = 64 lanes x (FB / 64) floats of one frame row
    constexpr int RV = FB / 64;  // floats per lane (1 or 2)
    for (int s0 = 0; s0 < w; s0 += 16 * 8) {  // 16 waves x 8 frames per round
        float v[8][RV];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int t = x0 + s0 + wv + 16 * i;
            const int f = f0 + lane * RV;
            const bool ok = s0 + wv + 16 * i < w && f + RV - 1 < HP;
            if (RV == 1) v[i][0] = ok ? sp[(size_t)t * HP + f] : 0.f;
            if (RV == 2) {
                const float2 x = ok ? *reinterpret_cast<const float2 *>(sp + (size_t)t * HP + f) : make_float2(0, 0);
                v[i][0] = x.x;
                v[i][RV - 1] = x.y;
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int tt = s0 + wv + 16 * i;
            if (tt < w)
#pragma unroll
                for (int j = 0; j < RV; j++) tile[(lane * RV + j) * PITCH + tt] = (uint16_t)(int)v[i][j];
        }
    }
    __syncthreads();
    // write u16 rows: the tile column's CORE frames [512 tx, 512 tx + core) of rows f0 .. f0 + FB (4-byte stores, 2 frames per lane)
    uint16_t *im = img + (size_t)n * H * TP;
    const int cs = tx * 512 - x0, core = (T - tx * 512 < 512 ? T - tx * 512 : 512);
    for (int r = wv; r < FB && f0 + r < H; r += 16) {
        for (int c = 2 * lane; c < core; c += 128) {
            const uint32_t p = tile[r * PITCH + cs + c] | ((uint32_t)tile[r * PITCH + cs + c + 1] << 16);
            *reinterpret_cast<uint32_t *>(im + (size_t)(f0 + r) * TP + tx * 512 + c) = p;
        }
    }
    // write RGBA rows: row f of the image -> every tile row (tx, ty) that contains it (own tile + gutters of the neighbours)
    uint8_t *rg = rgba + (size_t)n * G.total;
    for (int r = wv; r < FB && f0 + r < H; r += 16) {
        const int f = f0 + r;
#pragma unroll
        for (int ty = 0; ty < NTY; ty++) {
            if (f < G.oy[ty] || f >= G.oy[ty] + G.h[ty]) continue;
            uint32_t *dst = reinterpret_cast<uint32_t *>(rg + G.base[tx][ty]) + (size_t)(G.oy[ty] + G.h[ty] - 1 - f) * w;
            // 16-byte stores on the destination's 16-byte grid (row starts are only 4-byte aligned for odd widths)
            const int mis = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
            for (int q = lane; 4 * q - mis < w; q += 64) {
                const int c = 4 * q - mis;
                if (c >= 0 && c + 4 <= w) {
                    uint4 o;
                    o.x = lutc(lut, tile[r * PITCH + c]);
                    o.y = lutc(lut, tile[r * PITCH + c + 1]);
                    o.z = lutc(lut, tile[r * PITCH + c + 2]);
                    o.w = lutc(lut, tile[r * PITCH + c + 3]);
                    *reinterpret_cast<uint4 *>(dst + c) = o;
                } else {
                    for (int k = (c < 0 ? 0 : c); k < c + 4 && k < w; k++) dst[k] = lutc(lut, tile[r * PITCH + k]);
                }
            }
        }
    }
}

// ---- mode 3: as mode 2 with 128 freq rows, engineered for bytes in flight: every lane requests its whole share of the block's
// 520 x 512 B of spec (16-byte loads: 32 lanes per row piece, 2 frames per wave-instruction, 17 requests per lane) before it
// converts any of it; the u16 rows leave as one 1 KB row per wave-instruction (16 B per lane), the RGBA rows as 1 KB pieces
// (16 B = 4 px per lane), both from 16- / 8-byte LDS reads.  One block per CU (133 KB of LDS), 1024 threads.
template <int FB, int THREADS>
__global__ __launch_bounds__(THREADS) void fused_cols_deep(const float *__restrict__ spec, uint16_t *__restrict__ img, uint8_t *__restrict__ rgba,
                                                           const uint32_t *__restrict__ cmap) {
    constexpr int PITCH = 536, WAVES = THREADS / 64;  // u16 per LDS row (1072 B: 16-byte aligned rows)
    constexpr int LPR = FB / 4;                       // lanes per 16-byte-per-lane row piece
    constexpr int FPI = 64 / LPR;                     // frames per wave-instruction
    constexpr int NLD = (520 + FPI * WAVES - 1) / (FPI * WAVES);
    extern __shared__ __attribute__((aligned(16))) uint16_t tile[];
    __shared__ uint32_t lut[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 256) lut[tid] = cmap[tid];
    constexpr int BANDS = (H + FB - 1) / FB;
    const int b = blockIdx.x, n = b / (BANDS * NTX), l = b % (BANDS * NTX);
    const int band = l % BANDS, tx = l / BANDS;
    const int f0 = band * FB, x0 = G.ox[tx], w = G.w[tx];
    const float *sp = spec + (size_t)n * T * HP;
    const int fl = 4 * (lane % LPR), fr = lane / LPR;
    float4 v[NLD];
#pragma unroll
    for (int i = 0; i < NLD; i++) {
        const int tt = FPI * (wv + WAVES * i) + fr;
        const bool ok = tt < w && f0 + fl + 3 < HP;
        v[i] = ok ? *reinterpret_cast<const float4 *>(sp + (size_t)(x0 + tt) * HP + f0 + fl) : make_float4(0, 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NLD; i++) {
        const int tt = FPI * (wv + WAVES * i) + fr;
        if (tt < w) {
            tile[(fl + 0) * PITCH + tt] = (uint16_t)(int)v[i].x;
            tile[(fl + 1) * PITCH + tt] = (uint16_t)(int)v[i].y;
            tile[(fl + 2) * PITCH + tt] = (uint16_t)(int)v[i].z;
            tile[(fl + 3) * PITCH + tt] = (uint16_t)(int)v[i].w;
        }
    }
    __syncthreads();
    uint16_t *im = img + (size_t)n * H * TP;
    const int cs = tx * 512 - x0, core = (T - tx * 512 < 512 ? T - tx * 512 : 512);
#pragma unroll
    for (int i = 0; i < (FB + WAVES - 1) / WAVES; i++) {
        const int r = wv + WAVES * i, c = 8 * lane;
        if (r < FB && f0 + r < H && c < core) {
            const uint2 a = *reinterpret_cast<const uint2 *>(&tile[r * PITCH + cs + c]);
            const uint2 bq = *reinterpret_cast<const uint2 *>(&tile[r * PITCH + cs + c + 4]);
            *reinterpret_cast<uint4 *>(im + (size_t)(f0 + r) * TP + tx * 512 + c) = make_uint4(a.x, a.y, bq.x, bq.y);
        }
    }
    uint8_t *rg = rgba + (size_t)n * G.total;
#pragma unroll 1
    for (int i = 0; i < (FB + WAVES - 1) / WAVES; i++) {
        const int r = wv + WAVES * i, f = f0 + r;
        if (r >= FB || f >= H) continue;
#pragma unroll
        for (int ty = 0; ty < NTY; ty++) {
            if (f < G.oy[ty] || f >= G.oy[ty] + G.h[ty]) continue;
            uint32_t *dst = reinterpret_cast<uint32_t *>(rg + G.base[tx][ty]) + (size_t)(G.oy[ty] + G.h[ty] - 1 - f) * w;
            const int mis = (int)((reinterpret_cast<uintptr_t>(dst) >> 2) & 3);
#pragma unroll
            for (int k = 0; k < 3; k++) {
                const int c = 4 * (lane + 64 * k) - mis;
                if (c >= w) continue;
                if (c >= 0 && c + 4 <= w) {
                    const uint16_t *s4 = &tile[r * PITCH + c];
                    *reinterpret_cast<uint4 *>(dst + c) = make_uint4(lutc(lut, s4[0]), lutc(lut, s4[1]), lutc(lut, s4[2]), lutc(lut, s4[3]));
                } else {
                    for (int q = (c < 0 ? 0 : c); q < c + 4 && q < w; q++) dst[q] = lutc(lut, tile[r * PITCH + q]);
                }
            }
        }
    }
}

// ---- mode 1: block = 128 freq x 64 frames, RGBA pieces straight from the block
__global__ __launch_bounds__(256) void fused_small(const float *__restrict__ spec, uint16_t *__restrict__ img, uint8_t *__restrict__ rgba,
                                                   const uint32_t *__restrict__ cmap, int write_rgba) {
    constexpr int TF = 128, TT = 64, PITCH = TT + 2;
    __shared__ uint16_t tile[TF * PITCH];
    __shared__ uint32_t lut[256];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    lut[tid] = cmap[tid];
    constexpr int tiles_f = (H + TF - 1) / TF, tiles_t = (T + TT - 1) / TT;
    const int b = blockIdx.x, per = tiles_f * tiles_t, n = b / per, l = b % per;
    const int f0 = (l % tiles_f) * TF, t0 = (l / tiles_f) * TT;
    const float *sp = spec + (size_t)n * T * HP;
    float v[16][2];
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int t = t0 + wv + 4 * i, f = f0 + 2 * lane;
        const float2 x = (t < T && f + 1 < HP) ? *reinterpret_cast<const float2 *>(sp + (size_t)t * HP + f) : make_float2(0, 0);
        v[i][0] = x.x;
        v[i][1] = x.y;
    }
#pragma unroll
    for (int i = 0; i < 16; i++) {
        tile[(2 * lane) * PITCH + wv + 4 * i] = (uint16_t)(int)v[i][0];
        tile[(2 * lane + 1) * PITCH + wv + 4 * i] = (uint16_t)(int)v[i][1];
    }
    __syncthreads();
    uint16_t *im = img + (size_t)n * H * TP;
    const int half = lane >> 5, tl = 2 * (lane & 31);
#pragma unroll
    for (int i = 0; i < 16; i++) {
        const int r = 2 * (wv + 4 * i) + half, f = f0 + r, t = t0 + tl;
        if (f < H && t + 1 < TP) *reinterpret_cast<uint32_t *>(im + (size_t)f * TP + t) = *reinterpret_cast<const uint32_t *>(&tile[r * PITCH + tl]);
    }
    if (!write_rgba) return;
    uint8_t *rg = rgba + (size_t)n * G.total;
    // 16 lanes x 4 px = one 64-frame piece of a row; a wave-instruction covers 4 rows
    const int rr = lane >> 4, c4 = 4 * (lane & 15);
    for (int i = 0; i < 8; i++) {
        const int r = 4 * (wv + 4 * i) + rr, f = f0 + r;
        if (f >= H) continue;
        for (int tx = 0; tx < NTX; tx++) {
            if (t0 + c4 + 3 < G.ox[tx] || t0 + c4 >= G.ox[tx] + G.w[tx]) continue;
            for (int ty = 0; ty < NTY; ty++) {
                if (f < G.oy[ty] || f >= G.oy[ty] + G.h[ty]) continue;
                uint32_t *dst = reinterpret_cast<uint32_t *>(rg + G.base[tx][ty]) + (size_t)(G.oy[ty] + G.h[ty] - 1 - f) * G.w[tx];
                for (int k = 0; k < 4; k++) {
                    const int x = t0 + c4 + k - G.ox[tx];
                    if (x >= 0 && x < G.w[tx] && t0 + c4 + k < TP) dst[x] = lutc(lut, tile[r * PITCH + c4 + k]);
                }
            }
        }
    }
}

// ---- mode 0, second half: linear raster of the u16 image into the tiles (the product kernel's structure: 16-byte stores along rows)
__global__ __launch_bounds__(256) void raster_lin(const uint16_t *__restrict__ img, uint8_t *__restrict__ rgba, const uint32_t *__restrict__ cmap) {
    __shared__ uint32_t lut[256];
    lut[threadIdx.x] = cmap[threadIdx.x];
    __syncthreads();
    const int n = blockIdx.y, tile = blockIdx.z, tx = tile / NTY, ty = tile % NTY;
    const int w = G.w[tx], h = G.h[ty];
    const uint16_t *im = img + (size_t)n * H * TP;
    uint32_t *dst = reinterpret_cast<uint32_t *>(rgba + (size_t)n * G.total + G.base[tx][ty]);
    const int nq = (w * h + 3) / 4;
    for (int q = blockIdx.x * 256 + threadIdx.x; q < nq; q += gridDim.x * 256) {
        uint32_t o[4];
        for (int k = 0; k < 4; k++) {
            const int p = 4 * q + k, r = p / w, c = p - r * w;
            o[k] = p < w * h ? lutc(lut, im[(size_t)(G.oy[ty] + h - 1 - r) * TP + G.ox[tx] + c]) : 0;
        }
        if (4 * q + 4 <= w * h) *reinterpret_cast<uint4 *>(dst + 4 * q) = make_uint4(o[0], o[1], o[2], o[3]);
        else for (int k = 0; 4 * q + k < w * h; k++) dst[4 * q + k] = o[k];
    }
}

template <class F>
static float time_ms(F f) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    float sum = 0;
    for (int i = 0; i < 10; i++) {
        hipEventRecord(e0);
        f();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    return sum / 10;
}

int main() {
    const Geom g = make_geom();
    hipMemcpyToSymbol(HIP_SYMBOL(G), &g, sizeof g);
    float *a;
    uint16_t *b;
    uint8_t *c;
    uint32_t *cm;
    hipMalloc(&a, (size_t)N * T * HP * 4);
    hipMalloc(&b, (size_t)N * H * TP * 2);
    hipMalloc(&c, (size_t)N * g.total);
    hipMalloc(&cm, 1024);
    hipMemset(a, 0, (size_t)N * T * HP * 4);
    hipMemset(cm, 0, 1024);
    const double px = (double)N * T * H;
    {
        const int blocks = ((H + 127) / 128) * ((T + 63) / 64) * N;
        const float q = time_ms([&] { hipLaunchKernelGGL(fused_small, dim3(blocks), dim3(256), 0, 0, a, b, c, cm, 0); });
        const float r = time_ms([&] { hipLaunchKernelGGL(raster_lin, dim3(8, N, NTX * NTY), dim3(256), 0, 0, b, c, cm); });
        printf("mode 0  quantise 128x64 %.3f ms + linear raster %.3f ms = %.3f ms   (%.0f GB/s of 12 B/px)\n", q, r, q + r, px * 12 / (q + r) / 1e6);
        const float f = time_ms([&] { hipLaunchKernelGGL(fused_small, dim3(blocks), dim3(256), 0, 0, a, b, c, cm, 1); });
        printf("mode 1  fused, 128 freq x 64 frames, 256-byte RGBA pieces: %.3f ms   (%.0f GB/s of 10 B/px)\n", f, px * 10 / f / 1e6);
    }
    {
        constexpr int FB = 128;
        const size_t lds = (size_t)FB * 530 * 2;
        hipFuncSetAttribute(reinterpret_cast<const void *>(fused_cols<FB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int blocks = ((H + FB - 1) / FB) * NTX * N;
        const float f = time_ms([&] { hipLaunchKernelGGL(fused_cols<FB>, dim3(blocks), dim3(1024), lds, 0, a, b, c, cm); });
        printf("mode 2  fused, %d freq x tile column in LDS (%.0f KB), whole rows: %.3f ms   (%.0f GB/s of 10 B/px)\n", FB, lds / 1024.0, f, px * 10 / f / 1e6);
    }
    {
        constexpr int FB = 64;
        const size_t lds = (size_t)FB * 530 * 2;
        hipFuncSetAttribute(reinterpret_cast<const void *>(fused_cols<FB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        const int blocks = ((H + FB - 1) / FB) * NTX * N;
        const float f = time_ms([&] { hipLaunchKernelGGL(fused_cols<FB>, dim3(blocks), dim3(1024), lds, 0, a, b, c, cm); });
        printf("mode 2  fused, %d freq x tile column in LDS (%.0f KB), whole rows: %.3f ms   (%.0f GB/s of 10 B/px)\n", FB, lds / 1024.0, f, px * 10 / f / 1e6);
    }
#define RUN_DEEP(FB, TH)                                                                                                    \
    {                                                                                                                        \
        const size_t lds = (size_t)(FB) * 536 * 2;                                                                           \
        hipFuncSetAttribute(reinterpret_cast<const void *>(fused_cols_deep<FB, TH>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        const int blocks = ((H + (FB) - 1) / (FB)) * NTX * N;                                                                \
        const float f = time_ms([&] { hipLaunchKernelGGL((fused_cols_deep<FB, TH>), dim3(blocks), dim3(TH), lds, 0, a, b, c, cm); }); \
        printf("mode 3  fused, %3d freq x tile column in LDS (%.0f KB), %4d threads, all loads of a lane in flight, whole rows out: %.3f ms   (%.0f GB/s of 10 B/px)\n", \
               FB, lds / 1024.0, TH, f, px * 10 / f / 1e6);                                                                   \
    }
    RUN_DEEP(128, 1024)
    RUN_DEEP(64, 1024)
    RUN_DEEP(64, 512)
    RUN_DEEP(32, 512)
    RUN_DEEP(32, 256)
    {
        const Geom ga = make_geom(true);
        hipMemcpyToSymbol(HIP_SYMBOL(G), &ga, sizeof ga);
        const int blocks = ((H + 127) / 128) * ((T + 63) / 64) * N;
        const float f = time_ms([&] { hipLaunchKernelGGL(fused_small, dim3(blocks), dim3(256), 0, 0, a, b, c, cm, 1); });
        printf("mode 1a fused, 128 freq x 64 frames, ALIGNED 256-byte RGBA pieces (no gutters, widths %% 64 == 0): %.3f ms   (%.0f GB/s of 10 B/px)\n", f, px * 10 / f / 1e6);
    }
    return 0;
}
