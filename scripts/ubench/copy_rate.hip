// Micro-benchmark (development tool): achievable HBM streaming rates on MI355X for the read:write mixes of
// the image kernels (2:1 quantise, 1:2 rasterise, 1:1 copy, read-only, write-only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

// each thread: read RB bytes (as 4/8/16-byte vectors), write WB bytes
template <int RV, int WV>
__global__ __launch_bounds__(256) void k(const uint32_t *__restrict__ src, uint32_t *__restrict__ dst, size_t n_threads_total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_threads_total) return;
    uint32_t acc = 0;
    if (RV > 0) {
        uint32_t v[RV > 0 ? RV : 1];
        if (RV == 4) { const uint4 t = reinterpret_cast<const uint4 *>(src)[i]; v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
        if (RV == 2) { const uint2 t = reinterpret_cast<const uint2 *>(src)[i]; v[0] = t.x; v[1] = t.y; }
        if (RV == 1) { v[0] = src[i]; }
        for (int j = 0; j < RV; j++) acc += v[j];
    }
    if (WV == 4) reinterpret_cast<uint4 *>(dst)[i] = make_uint4(acc, acc + 1, acc + 2, acc + 3);
    if (WV == 2) reinterpret_cast<uint2 *>(dst)[i] = make_uint2(acc, acc + 1);
    if (WV == 1) dst[i] = acc;
    if (WV == 0 && acc == 0x12345678u) dst[0] = acc;
}

template <int RV, int WV>
void run(const char *name, uint32_t *a, uint32_t *b, size_t n_threads) {
    const int blocks = (int)((n_threads + 255) / 256);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<RV, WV>), dim3(blocks), dim3(256), 0, 0, a, b, n_threads);
    float best = 1e9, sum = 0;
    const int reps = 10;
    for (int i = 0; i < reps; i++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<RV, WV>), dim3(blocks), dim3(256), 0, 0, a, b, n_threads);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
        sum += ms;
    }
    const double bytes = (double)n_threads * 4.0 * (RV + WV);
    printf("%-34s %6.2f GB  avg %.3f ms  best %.3f ms  -> %.0f GB/s avg, %.0f GB/s best\n", name, bytes / 1e9, sum / reps, best,
           bytes / (sum / reps) / 1e6, bytes / best / 1e6);
}

int main() {
    const size_t cap = (size_t)3 << 30;  // 3 GiB each
    uint32_t *a, *b;
    hipMalloc(&a, cap);
    hipMalloc(&b, cap);
    hipMemset(a, 1, cap);
    hipMemset(b, 0, cap);
    const size_t n = (size_t)96 << 20;  // threads
    run<4, 0>("read 16 B/thread", a, b, n);
    run<0, 4>("write 16 B/thread", a, b, n);
    run<4, 4>("copy 16 B -> 16 B", a, b, n);
    run<2, 4>("read 8 B -> write 16 B (raster mix)", a, b, n);
    run<4, 2>("read 16 B -> write 8 B (quantise mix)", a, b, n);
    run<1, 1>("copy 4 B -> 4 B", a, b, n);
    run<2, 2>("copy 8 B -> 8 B", a, b, n);
    return 0;
}
