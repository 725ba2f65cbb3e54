// Micro-benchmark (development tool): issue rate of a few VALU instruction kinds on gfx950 as a
// function of waves per SIMD.  Prints cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP 64
template <int KIND>
__global__ void k(float *out, int iters) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float b = 1.0001f, c = 0.5f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7};
    const v2 pb = {b, b}, pc = {c, c};
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < REP / 8; r++) {
            if (KIND == 0) {  // v_fma_f32 x8 independent
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));
            } else if (KIND == 1) {  // v_pk_fma_f32 x4 (= 8 fma)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb), "v"(pc));
            } else if (KIND == 2) {  // v_add_f32
                asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                             "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
            } else if (KIND == 3) {  // v_log_f32
                asm volatile("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n"
                             "v_log_f32 %4, %4\n v_log_f32 %5, %5\n v_log_f32 %6, %6\n v_log_f32 %7, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 4) {  // v_permlane32_swap
                asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 5) {  // v_permlane16_swap
                asm volatile("v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            } else if (KIND == 6) {  // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             "v_pk_add_f32 %0, %0, %4\n v_pk_add_f32 %1, %1, %4\n v_pk_add_f32 %2, %2, %4\n v_pk_add_f32 %3, %3, %4\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pb));
            } else if (KIND == 7) {  // v_mov_b32 dpp row_mirror
                asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %4, %5 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %7 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %1, %0 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %2 row_mirror row_mask:0xf bank_mask:0xf\n"
                             "v_mov_b32_dpp %5, %4 row_mirror row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %6 row_mirror row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

template <int KIND>
void run(const char *name, int per_instr_count) {
    float *d;
    hipMalloc(&d, 256 * 1024 * 16 * sizeof(float));
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {  // waves per SIMD: block = 256 threads (1 wave per SIMD), wps blocks per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 10);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr_per_wave = (double)iters * REP * per_instr_count / 8.0;  // wave-instructions per wave
        const double ns_per_instr_per_simd = ms * 1e6 / (instr_per_wave * wps);
        printf("%-20s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, wps, ms,
               ns_per_instr_per_simd, ns_per_instr_per_simd * 2.4);
    }
    hipFree(d);
}

int main() {
    run<0>("v_fma_f32", 8);
    run<2>("v_add_f32", 8);
    run<1>("v_pk_fma_f32", 8);
    run<6>("v_pk_add_f32", 8);
    run<3>("v_log_f32", 8);
    run<4>("v_permlane32_swap", 8);
    run<5>("v_permlane16_swap", 8);
    run<7>("v_mov_dpp row_mirror", 8);
    return 0;
}
