// Micro-benchmark (development tool): does the VGPR bank of the source operands change the VALU issue
// rate on gfx950?  4 waves per SIMD; prints cycles (s_memtime) per wave-instruction per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

// KIND 0: v_fma d, a, b, c with a,b,c in the same bank (reg index mod 4 equal)
//      1: v_fma with a,b,c in three different banks
//      2: v_add_f32 d, a, b same bank      3: v_add_f32 different banks
//      4: v_fma d, a, b, d (accumulate; 2 distinct sources + dst)   5: v_mul_f32 d, literal, a
//      6: v_fmac_f32 d, a, b               7: v_mov_b32           8: v_fma d, s, a, b (SGPR operand)
template <int KIND>
__global__ __launch_bounds__(1024) void k(float *out, long long *cyc, int iters) {
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
        if (KIND == 0)
            asm volatile(
                "v_fma_f32 v40, v4, v8, v12\n v_fma_f32 v41, v5, v9, v13\n v_fma_f32 v42, v6, v10, v14\n v_fma_f32 v43, v7, v11, v15\n"
                "v_fma_f32 v44, v16, v20, v24\n v_fma_f32 v45, v17, v21, v25\n v_fma_f32 v46, v18, v22, v26\n v_fma_f32 v47, v19, v23, v27\n"
                "v_fma_f32 v48, v4, v8, v12\n v_fma_f32 v49, v5, v9, v13\n v_fma_f32 v50, v6, v10, v14\n v_fma_f32 v51, v7, v11, v15\n"
                "v_fma_f32 v52, v16, v20, v24\n v_fma_f32 v53, v17, v21, v25\n v_fma_f32 v54, v18, v22, v26\n v_fma_f32 v55, v19, v23, v27\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 1)
            asm volatile(
                "v_fma_f32 v40, v4, v9, v14\n v_fma_f32 v41, v5, v10, v15\n v_fma_f32 v42, v6, v11, v12\n v_fma_f32 v43, v7, v8, v13\n"
                "v_fma_f32 v44, v16, v21, v26\n v_fma_f32 v45, v17, v22, v27\n v_fma_f32 v46, v18, v23, v24\n v_fma_f32 v47, v19, v20, v25\n"
                "v_fma_f32 v48, v4, v9, v14\n v_fma_f32 v49, v5, v10, v15\n v_fma_f32 v50, v6, v11, v12\n v_fma_f32 v51, v7, v8, v13\n"
                "v_fma_f32 v52, v16, v21, v26\n v_fma_f32 v53, v17, v22, v27\n v_fma_f32 v54, v18, v23, v24\n v_fma_f32 v55, v19, v20, v25\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 2)
            asm volatile(
                "v_add_f32 v40, v4, v8\n v_add_f32 v41, v5, v9\n v_add_f32 v42, v6, v10\n v_add_f32 v43, v7, v11\n"
                "v_add_f32 v44, v16, v20\n v_add_f32 v45, v17, v21\n v_add_f32 v46, v18, v22\n v_add_f32 v47, v19, v23\n"
                "v_add_f32 v48, v4, v8\n v_add_f32 v49, v5, v9\n v_add_f32 v50, v6, v10\n v_add_f32 v51, v7, v11\n"
                "v_add_f32 v52, v16, v20\n v_add_f32 v53, v17, v21\n v_add_f32 v54, v18, v22\n v_add_f32 v55, v19, v23\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 3)
            asm volatile(
                "v_add_f32 v40, v4, v9\n v_add_f32 v41, v5, v10\n v_add_f32 v42, v6, v11\n v_add_f32 v43, v7, v8\n"
                "v_add_f32 v44, v16, v21\n v_add_f32 v45, v17, v22\n v_add_f32 v46, v18, v23\n v_add_f32 v47, v19, v20\n"
                "v_add_f32 v48, v4, v9\n v_add_f32 v49, v5, v10\n v_add_f32 v50, v6, v11\n v_add_f32 v51, v7, v8\n"
                "v_add_f32 v52, v16, v21\n v_add_f32 v53, v17, v22\n v_add_f32 v54, v18, v23\n v_add_f32 v55, v19, v20\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 4)
            asm volatile(
                "v_fma_f32 v40, v4, v9, v40\n v_fma_f32 v41, v5, v10, v41\n v_fma_f32 v42, v6, v11, v42\n v_fma_f32 v43, v7, v8, v43\n"
                "v_fma_f32 v44, v16, v21, v44\n v_fma_f32 v45, v17, v22, v45\n v_fma_f32 v46, v18, v23, v46\n v_fma_f32 v47, v19, v20, v47\n"
                "v_fma_f32 v48, v4, v9, v48\n v_fma_f32 v49, v5, v10, v49\n v_fma_f32 v50, v6, v11, v50\n v_fma_f32 v51, v7, v8, v51\n"
                "v_fma_f32 v52, v16, v21, v52\n v_fma_f32 v53, v17, v22, v53\n v_fma_f32 v54, v18, v23, v54\n v_fma_f32 v55, v19, v20, v55\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 5)
            asm volatile(
                "v_mul_f32 v40, 0x3f6c835e, v4\n v_mul_f32 v41, 0x3f6c835e, v5\n v_mul_f32 v42, 0x3f6c835e, v6\n v_mul_f32 v43, 0x3f6c835e, v7\n"
                "v_mul_f32 v44, 0x3f6c835e, v16\n v_mul_f32 v45, 0x3f6c835e, v17\n v_mul_f32 v46, 0x3f6c835e, v18\n v_mul_f32 v47, 0x3f6c835e, v19\n"
                "v_mul_f32 v48, 0x3f6c835e, v4\n v_mul_f32 v49, 0x3f6c835e, v5\n v_mul_f32 v50, 0x3f6c835e, v6\n v_mul_f32 v51, 0x3f6c835e, v7\n"
                "v_mul_f32 v52, 0x3f6c835e, v16\n v_mul_f32 v53, 0x3f6c835e, v17\n v_mul_f32 v54, 0x3f6c835e, v18\n v_mul_f32 v55, 0x3f6c835e, v19\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 6)
            asm volatile(
                "v_fmac_f32 v40, v4, v9\n v_fmac_f32 v41, v5, v10\n v_fmac_f32 v42, v6, v11\n v_fmac_f32 v43, v7, v8\n"
                "v_fmac_f32 v44, v16, v21\n v_fmac_f32 v45, v17, v22\n v_fmac_f32 v46, v18, v23\n v_fmac_f32 v47, v19, v20\n"
                "v_fmac_f32 v48, v4, v9\n v_fmac_f32 v49, v5, v10\n v_fmac_f32 v50, v6, v11\n v_fmac_f32 v51, v7, v8\n"
                "v_fmac_f32 v52, v16, v21\n v_fmac_f32 v53, v17, v22\n v_fmac_f32 v54, v18, v23\n v_fmac_f32 v55, v19, v20\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 7)
            asm volatile(
                "v_mov_b32 v40, v4\n v_mov_b32 v41, v5\n v_mov_b32 v42, v6\n v_mov_b32 v43, v7\n"
                "v_mov_b32 v44, v16\n v_mov_b32 v45, v17\n v_mov_b32 v46, v18\n v_mov_b32 v47, v19\n"
                "v_mov_b32 v48, v4\n v_mov_b32 v49, v5\n v_mov_b32 v50, v6\n v_mov_b32 v51, v7\n"
                "v_mov_b32 v52, v16\n v_mov_b32 v53, v17\n v_mov_b32 v54, v18\n v_mov_b32 v55, v19\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
        if (KIND == 8)
            asm volatile(
                "v_fma_f32 v40, s4, v9, v14\n v_fma_f32 v41, s4, v10, v15\n v_fma_f32 v42, s4, v11, v12\n v_fma_f32 v43, s4, v8, v13\n"
                "v_fma_f32 v44, s4, v21, v26\n v_fma_f32 v45, s4, v22, v27\n v_fma_f32 v46, s4, v23, v24\n v_fma_f32 v47, s4, v20, v25\n"
                "v_fma_f32 v48, s4, v9, v14\n v_fma_f32 v49, s4, v10, v15\n v_fma_f32 v50, s4, v11, v12\n v_fma_f32 v51, s4, v8, v13\n"
                "v_fma_f32 v52, s4, v21, v26\n v_fma_f32 v53, s4, v22, v27\n v_fma_f32 v54, s4, v23, v24\n v_fma_f32 v55, s4, v20, v25\n" ::
                    : "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55");
    }
    const long long t1 = clock64();
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = 0.f;
}

template <int KIND>
void run(const char *name) {
    float *d;
    long long *dc;
    const int blocks = 256, iters = 20000;
    hipMalloc(&d, blocks * 1024 * sizeof(float));
    hipMalloc(&dc, blocks * 16 * sizeof(long long));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, dc, 10);
    hipDeviceSynchronize();
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(1024), 0, 0, d, dc, iters);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks * 16);
    hipMemcpy(h.data(), dc, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double avg = 0;
    for (auto v : h) avg += v;
    avg /= h.size();
    // 4 waves per SIMD, 16 instructions per iteration per wave
    printf("%-44s %.2f cycles per wave-instruction per SIMD\n", name, avg / ((double)iters * 16 * 4));
    hipFree(d);
    hipFree(dc);
}

int main() {
    run<0>("v_fma 3 sources same bank");
    run<1>("v_fma 3 sources different banks");
    run<2>("v_add 2 sources same bank");
    run<3>("v_add 2 sources different banks");
    run<4>("v_fma d = a*b + d");
    run<5>("v_mul literal");
    run<6>("v_fmac");
    run<7>("v_mov");
    run<8>("v_fma with an SGPR source");
    return 0;
}
