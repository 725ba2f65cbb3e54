// Micro-benchmark (development tool): the quantise kernel's access pattern (f32 [T][Hp] -> u16 [H][Tp], transposed
// through LDS) without its arithmetic, for several tile shapes and access widths.  Which shape streams best?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int T = 2813, H = 1025, HP = 1056, TP = 2816, N = 128;

// TF freq rows x TT frames per block; RV floats per lane per load (along freq); WB bytes per lane per store (along time)
template <int TF, int TT, int RV, int WB, int ORDER>
__global__ __launch_bounds__(256) void k(const float *__restrict__ spec, uint16_t *__restrict__ img) {
    constexpr int PITCH = TT + 2;  // u16; odd dword count
    extern __shared__ uint16_t tile[];  // [TF][PITCH]
    constexpr int tiles_f = (H + TF - 1) / TF, tiles_t = (T + TT - 1) / TT;
    const int b = blockIdx.x;
    const int per = tiles_f * tiles_t;
    const int n = b / per;
    int l = b % per;
    if (ORDER >= 2) {  // blocks b, b+8, ... share an XCD (and its L2): make THEM neighbours
        const int per_xcd = (per + 7) / 8;
        const int l2 = (l % 8) * per_xcd + l / 8;
        if (per % 8 == 0) l = l2;
    }
    int f0, t0;
    if (ORDER & 1) { f0 = (l / tiles_t) * TF; t0 = (l % tiles_t) * TT; }   // time-fastest
    else { f0 = (l % tiles_f) * TF; t0 = (l / tiles_f) * TT; }             // frequency-fastest
    const float *sp = spec + (size_t)n * T * HP;
    uint16_t *im = img + (size_t)n * H * TP;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    static_assert(TF % (64 * RV) == 0, "");
    constexpr int FSEG = TF / (64 * RV);        // row segments per frame
    constexpr int NLD = TT * FSEG / 4;          // loads per lane
    float v[NLD][RV];
#pragma unroll
    for (int i = 0; i < NLD; i++) {
        const int idx = wv + 4 * i, t = t0 + idx / FSEG, f = f0 + (idx % FSEG) * 64 * RV + lane * RV;
        const bool ok = t < T && f + RV - 1 < HP;
        if (RV == 1) v[i][0] = ok ? sp[(size_t)t * HP + f] : 0.f;
        if (RV == 2) { float2 x = ok ? *reinterpret_cast<const float2 *>(sp + (size_t)t * HP + f) : make_float2(0, 0); v[i][0] = x.x; v[i][1] = x.y; }
        if (RV == 4) { float4 x = ok ? *reinterpret_cast<const float4 *>(sp + (size_t)t * HP + f) : make_float4(0, 0, 0, 0); v[i][0] = x.x; v[i][1] = x.y; v[i][2] = x.z; v[i][3] = x.w; }
    }
#pragma unroll
    for (int i = 0; i < NLD; i++) {
        const int idx = wv + 4 * i, tt = idx / FSEG, ff = (idx % FSEG) * 64 * RV + lane * RV;
#pragma unroll
        for (int j = 0; j < RV; j++) tile[(ff + j) * PITCH + tt] = (uint16_t)(int)v[i][j];
    }
    __syncthreads();
    constexpr int PPL = WB / 2;                 // u16 per lane per store
    constexpr int LPR = TT / PPL;               // lanes per row
    constexpr int RPI = 64 / LPR > 0 ? 64 / LPR : 1;  // rows per wave-instruction
    static_assert(LPR <= 64, "");
    constexpr int NST = TF / (4 * RPI);
#pragma unroll
    for (int i = 0; i < NST; i++) {
        const int row = (wv + 4 * i) * RPI + lane / LPR, tt = (lane % LPR) * PPL;
        const int f = f0 + row, t = t0 + tt;
        if (f < H && t + PPL <= TP) {
            const uint16_t *src = &tile[row * PITCH + tt];
            uint16_t *dst = im + (size_t)f * TP + t;
            if (WB == 4) *reinterpret_cast<uint32_t *>(dst) = *reinterpret_cast<const uint32_t *>(src);
            if (WB == 8) { uint2 x; x.x = *reinterpret_cast<const uint32_t *>(src); x.y = *reinterpret_cast<const uint32_t *>(src + 2); *reinterpret_cast<uint2 *>(dst) = x; }
            if (WB == 16) { uint4 x; x.x = *reinterpret_cast<const uint32_t *>(src); x.y = *reinterpret_cast<const uint32_t *>(src + 2); x.z = *reinterpret_cast<const uint32_t *>(src + 4); x.w = *reinterpret_cast<const uint32_t *>(src + 6); *reinterpret_cast<uint4 *>(dst) = x; }
        }
    }
}

template <int TF, int TT, int RV, int WB, int ORDER>
void run(const float *a, uint16_t *b) {
    constexpr int tiles = ((H + TF - 1) / TF) * ((T + TT - 1) / TT) * N;
    const size_t lds = (size_t)TF * (TT + 2) * 2;
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<TF, TT, RV, WB, ORDER>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<TF, TT, RV, WB, ORDER>), dim3(tiles), dim3(256), lds, 0, a, b);
    float sum = 0;
    for (int i = 0; i < 10; i++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<TF, TT, RV, WB, ORDER>), dim3(tiles), dim3(256), lds, 0, a, b);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    const double bytes = (double)N * T * H * 6.0;
    printf("order %d tile %3d freq x %3d frames, read %2d B/lane, write %2d B/lane, LDS %5.1f KB: %.3f ms  %.0f GB/s\n", ORDER, TF, TT, 4 * RV, WB,
           lds / 1024.0, sum / 10, bytes / (sum / 10) / 1e6);
}

int main() {
    float *a;
    uint16_t *b;
    hipMalloc(&a, (size_t)N * T * HP * 4);
    hipMalloc(&b, (size_t)N * H * TP * 2);
    hipMemset(a, 0, (size_t)N * T * HP * 4);
    run<128, 64, 2, 4, 0>(a, b);
    run<128, 64, 2, 4, 1>(a, b);
    run<128, 64, 2, 4, 2>(a, b);
    run<128, 64, 2, 4, 3>(a, b);
    run<64, 128, 1, 4, 0>(a, b);
    run<64, 128, 1, 4, 1>(a, b);
    run<64, 128, 1, 4, 2>(a, b);
    run<64, 128, 1, 4, 3>(a, b);
    run<128, 128, 2, 4, 2>(a, b);
    run<128, 128, 2, 4, 3>(a, b);
    return 0;
}
