// Probe (development tool): what address does ds_write_addtid_b32 use?  LDS is pre-filled with a marker, every wave of a
// 4-wave workgroup writes lane-tagged values with M0 = its own base (some above 64 KB), then the whole LDS is dumped.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(uint32_t *out, uint32_t n_dw, uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) {
    extern __shared__ uint32_t lds[];
    for (uint32_t i = threadIdx.x; i < n_dw; i += 256) lds[i] = 0xdead0000u;
    __syncthreads();
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const uint32_t base = wave == 0 ? b0 : wave == 1 ? b1 : wave == 2 ? b2 : b3;
    const uint32_t val = 0x1000u * (wave + 1) + lane;
    asm volatile("s_mov_b32 m0, %1\n\tds_write_addtid_b32 %0 offset:16\n\ts_waitcnt lgkmcnt(0)" ::"v"(val), "s"(base) : "memory");
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < n_dw; i += 256) out[i] = lds[i];
}
int main() {
    const uint32_t n_dw = 160 * 1024 / 4;
    uint32_t *d;
    hipMalloc(&d, n_dw * 4);
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, n_dw * 4);
    const uint32_t b[4] = {0, 4096, 70000 & ~3u, 0x10000u + 8192};
    hipLaunchKernelGGL(probe, dim3(1), dim3(256), n_dw * 4, 0, d, n_dw, b[0], b[1], b[2], b[3]);
    std::vector<uint32_t> h(n_dw);
    hipMemcpy(h.data(), d, n_dw * 4, hipMemcpyDeviceToHost);
    printf("bases (bytes): %u %u %u %u; offset 16\n", b[0], b[1], b[2], b[3]);
    uint32_t runs = 0;
    for (uint32_t i = 0; i < n_dw && runs < 40; i++)
        if (h[i] != 0xdead0000u && (i == 0 || h[i - 1] == 0xdead0000u || (h[i] & ~0xfffu) != (h[i - 1] & ~0xfffu))) {
            uint32_t j = i;
            while (j + 1 < n_dw && h[j + 1] == h[j] + 1) j++;
            printf("  byte %6u: values 0x%x .. 0x%x (%u dwords)\n", i * 4, h[i], h[j], j - i + 1);
            runs++;
            i = j;
        }
    return 0;
}
