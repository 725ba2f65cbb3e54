// Micro-benchmark (development tool): LDS instruction throughput per CU on gfx950 with 16 waves
// per CU (one 1024-thread workgroup per CU), in shader-clock cycles (s_memtime) per wave-instruction.
// Also prints the effective shader clock (cycles / wall time).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 16
// KIND: 0 ds_write_b64 linear   1 ds_write_b128 linear  2 ds_read_b64 linear  3 ds_read_b128 linear
//       4 ds_write_b64 row-padded (lane stride 136 B: exchange-1 pattern)   5 ds_write_b32 linear
//       6 ds_read_b32 linear    7 ds_write2_b64 (two 8-B at +0 / +4096)     8 ds_read2_b64
//       9 ds_write_b128 lane stride 144 B (16 contiguous values per lane, 16-B aligned rows)
//      10 ds_read_b64 row-padded gather (lane stride 8 B + 8 B per 16 lanes)
//      11 mixed: 8 write_b64 + 8 read_b64 + 64 v_fma  (overlap test)
//      12 64 v_fma only
template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters) {
    extern __shared__ char smem[];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned slab = wave * 9216u;  // 9 KB per wave
    unsigned addr;
    if (KIND == 0 || KIND == 2 || KIND == 7 || KIND == 8 || KIND == 11) addr = slab + lane * 8u;
    else if (KIND == 1 || KIND == 3) addr = slab + lane * 16u;
    else if (KIND == 4) addr = slab + lane * 136u;
    else if (KIND == 9) addr = slab + lane * 144u;
    else if (KIND == 10) addr = slab + lane * 8u + (lane >> 4) * 8u;
    else addr = slab + lane * 4u;
    typedef float v2 __attribute__((ext_vector_type(2)));
    typedef float v4 __attribute__((ext_vector_type(4)));
    v2 d2 = {(float)lane, 1.f};
    v4 d4 = {(float)lane, 1.f, 2.f, 3.f};
    float d1 = lane;
    v2 r2 = {0, 0};
    v4 r4 = {0, 0, 0, 0};
    float r1 = 0;
    float a0 = lane, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    const float b = 1.0001f, c = 0.5f;
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP; r++) {
            if (KIND == 0) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(d2), "n"(r * 512));
            if (KIND == 1) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(d4), "n"((r & 7) * 1024));
            if (KIND == 2) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(addr), "n"(r * 512));
            if (KIND == 3) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r4) : "v"(addr), "n"((r & 7) * 1024));
            if (KIND == 4) asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(d2), "n"(r * 8));
            if (KIND == 5) asm volatile("ds_write_b32 %0, %1 offset:%2" ::"v"(addr), "v"(d1), "n"(r * 256));
            if (KIND == 6) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r1) : "v"(addr), "n"(r * 256));
            if (KIND == 7) asm volatile("ds_write2_b64 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "v"(d2), "v"(d2), "n"((r & 1) * 128), "n"((r & 1) * 128 + 64));
            if (KIND == 8) asm volatile("ds_read2_b64 %0, %1 offset0:%2 offset1:%3" : "=v"(r4) : "v"(addr), "n"((r & 1) * 128), "n"((r & 1) * 128 + 64));
            if (KIND == 9) asm volatile("ds_write_b128 %0, %1 offset:%2" ::"v"(addr), "v"(d4), "n"((r & 7) * 16));
            if (KIND == 10) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(addr), "n"(r * 544));
        }
        if (KIND == 11) {
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(d2), "n"(r * 512));
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
#pragma unroll
            for (int r = 0; r < 8; r++) {
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r2) : "v"(addr), "n"(r * 512));
                asm volatile("v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            }
        }
        if (KIND == 12) {
#pragma unroll
            for (int r = 0; r < 8; r++)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __syncthreads();
    const long long t1 = clock64();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r2.x + r2.y + r4.x + r4.y + r4.z + r4.w + r1 + a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int KIND>
void run(const char *name, int lds_per_iter) {
    float *d;
    long long *dc;
    const int blocks = 256;
    hipMalloc(&d, blocks * 1024 * sizeof(float));
    hipMalloc(&dc, blocks * sizeof(long long));
    hipFuncSetAttribute(reinterpret_cast<const void *>(k<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 16 * 9216);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 16 * 9216, 0, d, dc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 16 * 9216, 0, d, dc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), dc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= blocks;
    const double per_cu_instrs = (double)iters * lds_per_iter * 16;  // all 16 waves of the CU
    printf("%-44s %.3f ms  s_memtime ticks/iter/CU %.1f  -> %.2f ticks per LDS wave-instr per CU;  %.2f ns per instr\n", name, ms,
           avg / iters, avg / per_cu_instrs, ms * 1e6 / per_cu_instrs);
    hipFree(d);
    hipFree(dc);
}

int main() {
    run<12>("64 v_fma per wave (16 waves)", 64);
    run<0>("ds_write_b64 linear", REP);
    run<4>("ds_write_b64 lane stride 136 B", REP);
    run<1>("ds_write_b128 linear", REP);
    run<9>("ds_write_b128 lane stride 144 B", REP);
    run<5>("ds_write_b32 linear", REP);
    run<7>("ds_write2_b64", REP);
    run<2>("ds_read_b64 linear", REP);
    run<10>("ds_read_b64 row-padded", REP);
    run<3>("ds_read_b128 linear", REP);
    run<6>("ds_read_b32 linear", REP);
    run<8>("ds_read2_b64", REP);
    run<11>("8 wr_b64 + 8 rd_b64 + 64 v_fma", 16);
    return 0;
}
