// Micro-benchmark (development tool): the STFT kernel's OUTPUT pattern alone.  Persistent waves (12 per CU) each write
// rows of 1025 floats (pitch 1056), 32 consecutive rows per chunk, chunks from a global queue — no compute.
// MODE 0: the kernel's pattern: 8 aligned 256-byte dword stores (bins l + 64 j) + 8 mirrored ones (bins 1024 - l - 64 j) + 1
// MODE 1: 16 aligned dword stores + 1      MODE 2: 4 x 16-byte stores per lane (bins 4 l + 256 j) + 1 dword
// MODE 3: 2 x 16-byte + ... (8 B per lane x 8)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE, int CHUNK, int THREADS>
__global__ __launch_bounds__(THREADS) void k(float *out, uint32_t n_rows, uint32_t *queue) {
    const uint32_t lane = threadIdx.x & 63u;
    for (;;) {
        uint32_t c = 0;
        if (lane == 0) c = atomicAdd(queue, 1u);
        c = __builtin_amdgcn_readfirstlane(c);
        const uint32_t r0 = c * CHUNK;
        if (r0 >= n_rows) break;
        for (uint32_t r = r0; r < r0 + CHUNK && r < n_rows; r++) {
            float *row = out + (size_t)r * 1056;
            const float v = (float)r;
            if (MODE == 0) {
#pragma unroll
                for (int j = 0; j < 8; j++) row[lane + 64 * j] = v;
#pragma unroll
                for (int j = 0; j < 8; j++) row[1024 - lane - 64 * j] = v;
                if (lane == 0) row[512] = v;
            } else if (MODE == 1) {
#pragma unroll
                for (int j = 0; j < 16; j++) row[lane + 64 * j] = v;
                if (lane == 0) row[1024] = v;
            } else if (MODE == 2) {
#pragma unroll
                for (int j = 0; j < 4; j++) reinterpret_cast<float4 *>(row)[lane + 64 * j] = make_float4(v, v, v, v);
                if (lane == 0) row[1024] = v;
            } else {
#pragma unroll
                for (int j = 0; j < 8; j++) reinterpret_cast<float2 *>(row)[lane + 64 * j] = make_float2(v, v);
                if (lane == 0) row[1024] = v;
            }
            // a little spacing between rows, like the ~8000 cycles of FFT work per frame: none here (worst case burst)
        }
    }
}

// static partition: block b owns a contiguous range of rows; its waves take interleaved rows (w, w + W, ...), so the W
// rows being written by a CU at any time are adjacent in memory
template <int THREADS>
__global__ __launch_bounds__(THREADS) void k_il(float *out, uint32_t n_rows) {
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, W = THREADS / 64;
    const uint32_t per = (n_rows + gridDim.x - 1) / gridDim.x;
    const uint32_t r_lo = blockIdx.x * per, r_hi = min(n_rows, r_lo + per);
    for (uint32_t r = r_lo + wave; r < r_hi; r += W) {
        float *row = out + (size_t)r * 1056;
        const float v = (float)r;
#pragma unroll
        for (int j = 0; j < 4; j++) reinterpret_cast<float4 *>(row)[lane + 64 * j] = make_float4(v, v, v, v);
        if (lane == 0) row[1024] = v;
    }
}
template <int THREADS>
void run_il(float *out, uint32_t n_rows) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float sum = 0;
    for (int i = 0; i < 13; i++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_il<THREADS>), dim3(256), dim3(THREADS), 0, 0, out, n_rows);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) sum += ms;
    }
    const double bytes = (double)n_rows * 4100.0;
    printf("rows interleaved over the waves of a CU, %2d waves/CU: %.3f ms  %.0f GB/s\n", THREADS / 64, sum / 10, bytes / (sum / 10) / 1e6);
}

// non-persistent: one wave per row, 4 rows per block, blocks in row order
template <int TAIL, int PITCH>
__global__ __launch_bounds__(256) void k_flat(float *out, uint32_t n_rows, uint32_t rows_per_wave) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4 + (threadIdx.x >> 6);
    for (uint32_t i = 0; i < rows_per_wave; i++) {
        const uint32_t r = w * rows_per_wave + i;
        if (r >= n_rows) return;
        float *row = out + (size_t)r * PITCH;
        const float v = (float)r;
#pragma unroll
        for (int j = 0; j < 4; j++) reinterpret_cast<float4 *>(row)[lane + 64 * j] = make_float4(v, v, v, v);
        if (TAIL == 1 && lane == 0) row[1024] = v;      // the lone dword: a partial 128-byte line
        if (TAIL == 2 && lane < 32) row[1024 + lane] = v;  // the whole line (bin 1024 + the pitch padding)
        if (TAIL == 3 && lane < 16) row[1024 + lane] = v;  // 64 bytes
        if (TAIL == 4 && lane < 8) row[1024 + lane] = v;   // 32 bytes
    }
}
template <int TAIL, int PITCH>
void run_flat(float *out, uint32_t n_rows, uint32_t rpw) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float sum = 0;
    const uint32_t blocks = (n_rows + 4 * rpw - 1) / (4 * rpw);
    for (int i = 0; i < 13; i++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k_flat<TAIL, PITCH>), dim3(blocks), dim3(256), 0, 0, out, n_rows, rpw);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) sum += ms;
    }
    const double bytes = (double)n_rows * 4100.0;
    printf("non-persistent, %u rows per wave, tail mode %d, pitch %d: %.3f ms  %.0f GB/s (of 4100-byte rows)\n", rpw, TAIL, PITCH, sum / 10,
           bytes / (sum / 10) / 1e6);
}

template <int MODE, int CHUNK, int THREADS>
void run(const char *name, float *out, uint32_t n_rows, uint32_t *q) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float sum = 0;
    for (int i = 0; i < 13; i++) {
        hipMemsetAsync(q, 0, 4, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, CHUNK, THREADS>), dim3(256), dim3(THREADS), 0, 0, out, n_rows, q);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 3) sum += ms;
    }
    const double bytes = (double)n_rows * 4100.0;
    printf("%-40s chunk %3d rows, %2d waves/CU: %.3f ms  %.0f GB/s\n", name, CHUNK, THREADS / 64, sum / 10, bytes / (sum / 10) / 1e6);
}

int main() {
    const uint32_t n_rows = 360064;
    float *out;
    uint32_t *q;
    hipMalloc(&out, (size_t)n_rows * 1056 * 4);
    hipMalloc(&q, 4);
    run<0, 32, 768>("kernel pattern", out, n_rows, q);
    run<2, 32, 768>("4 x 16-byte stores", out, n_rows, q);
    run<2, 4, 768>("4 x 16-byte stores", out, n_rows, q);
    run<2, 8, 768>("4 x 16-byte stores", out, n_rows, q);
    run<2, 128, 768>("4 x 16-byte stores", out, n_rows, q);
    run<2, 32, 512>("4 x 16-byte stores", out, n_rows, q);
    run<2, 32, 1024>("4 x 16-byte stores", out, n_rows, q);
    run<2, 32, 256>("4 x 16-byte stores", out, n_rows, q);
    run_flat<1, 1056>(out, n_rows, 4);
    run_flat<0, 1056>(out, n_rows, 4);
    run_flat<2, 1056>(out, n_rows, 4);
    run_flat<3, 1056>(out, n_rows, 4);
    run_flat<4, 1056>(out, n_rows, 4);
    run_flat<0, 1024>(out, n_rows, 4);
    run_flat<1, 1025>(out, n_rows, 4);
    run_flat<1, 1028>(out, n_rows, 4);
    run_flat<1, 1040>(out, n_rows, 4);
    run_il<768>(out, n_rows);
    run_il<512>(out, n_rows);
    run_il<1024>(out, n_rows);
    run_il<256>(out, n_rows);
    return 0;
}
