// Micro-benchmark (development tool): which ingredient of the raster kernel costs bandwidth?  Starts from the plain
// "read 8 B -> write 16 B" stream (6.2 TB/s) and adds the ingredients one at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

struct Job { const uint16_t *img; uint32_t *out; uint32_t first_block, pad; uint32_t w[12]; };

// FEAT bit 0: 4 quads per thread (blocks of 1024 quads)   bit 1: dependent table loads in the prologue
//      bit 2: LDS LUT fill + barrier                       bit 3: 4 LUT lookups per quad (else arithmetic)
template <int FEAT>
__global__ __launch_bounds__(256) void k(const uint2 *__restrict__ src, uint4 *__restrict__ dst, const uint32_t *__restrict__ block_job,
                                         const Job *__restrict__ jobs, const uint32_t *__restrict__ colormap, size_t n_quads) {
    __shared__ uint32_t lut[1024];
    constexpr int NIT = (FEAT & 1) ? 4 : 1;
    size_t base = (size_t)blockIdx.x * 256 * NIT;
    if (FEAT & 2) {
        const Job job = jobs[block_job[blockIdx.x]];
        base = (size_t)(blockIdx.x - job.first_block) * 256 * NIT + (size_t)job.first_block * 256 * NIT;
        src = reinterpret_cast<const uint2 *>(job.img);
        dst = reinterpret_cast<uint4 *>(job.out);
    }
    if (FEAT & 4) {
        for (uint32_t i = threadIdx.x; i < 258; i += 256) lut[i] = colormap[i];
        __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < NIT; it++) {
        const size_t q = base + it * 256 + threadIdx.x;
        if (q >= n_quads) break;
        const uint2 w = src[q];
        uint32_t v[4] = {w.x & 0xffffu, w.x >> 16, w.y & 0xffffu, w.y >> 16};
        uint32_t p[4];
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (FEAT & 8) {
                const uint32_t x = v[i] * 257u + 32767u;
                p[i] = lut[(x + 1u + (x >> 16)) >> 16];
            } else {
                p[i] = v[i] * 0x10101u;
            }
        }
        dst[q] = make_uint4(p[0], p[1], p[2], p[3]);
    }
}

template <int FEAT>
void run(const char *name, uint2 *a, uint4 *b, uint32_t *bj, Job *jobs, uint32_t *cm, size_t n_quads) {
    constexpr int NIT = (FEAT & 1) ? 4 : 1;
    const int blocks = (int)((n_quads + 256 * NIT - 1) / (256 * NIT));
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) hipLaunchKernelGGL((k<FEAT>), dim3(blocks), dim3(256), 0, 0, a, b, bj, jobs, cm, n_quads);
    float sum = 0;
    const int reps = 10;
    for (int i = 0; i < reps; i++) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<FEAT>), dim3(blocks), dim3(256), 0, 0, a, b, bj, jobs, cm, n_quads);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        sum += ms;
    }
    const double bytes = (double)n_quads * 24.0;
    printf("%-60s avg %.3f ms  -> %.0f GB/s\n", name, sum / reps, bytes / (sum / reps) / 1e6);
}

int main() {
    const size_t n_quads = (size_t)92 << 20;  // ~ the bench workload (369 Mpx)
    uint2 *a;
    uint4 *b;
    hipMalloc(&a, n_quads * 8);
    hipMalloc(&b, n_quads * 16);
    hipMemset(a, 1, n_quads * 8);
    const int max_blocks = (int)((n_quads + 255) / 256);
    uint32_t *bj, *cm;
    Job *jobs;
    hipMalloc(&bj, max_blocks * 4);
    hipMemset(bj, 0, max_blocks * 4);  // every block -> job 0
    hipMalloc(&cm, 1024 * 4);
    hipMemset(cm, 7, 1024 * 4);
    Job hj{};
    hj.img = reinterpret_cast<const uint16_t *>(a);
    hj.out = reinterpret_cast<uint32_t *>(b);
    hj.first_block = 0;
    hipMalloc(&jobs, sizeof(Job));
    hipMemcpy(jobs, &hj, sizeof(Job), hipMemcpyHostToDevice);
    run<0>("plain: 1 quad/thread", a, b, bj, jobs, cm, n_quads);
    run<1>("4 quads/thread", a, b, bj, jobs, cm, n_quads);
    run<2>("1 quad/thread + dependent table loads", a, b, bj, jobs, cm, n_quads);
    run<3>("4 quads/thread + dependent table loads", a, b, bj, jobs, cm, n_quads);
    run<4>("1 quad/thread + LUT fill + barrier", a, b, bj, jobs, cm, n_quads);
    run<5>("4 quads/thread + LUT fill + barrier", a, b, bj, jobs, cm, n_quads);
    run<13>("4 quads/thread + LUT fill + lookups", a, b, bj, jobs, cm, n_quads);
    run<15>("4 quads/thread + table loads + LUT fill + lookups (= raster)", a, b, bj, jobs, cm, n_quads);
    run<14>("1 quad/thread + table loads + LUT fill + lookups", a, b, bj, jobs, cm, n_quads);
    return 0;
}
