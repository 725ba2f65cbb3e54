// Development probe (round 5): is v_cvt_rpi_i32_f32 (floor(x + 0.5)) EXACT on gfx950, i.e. equal to round-half-away for every
// non-negative f32 below 2^18 — in particular for 0.5 - 2^-25, where an f32 addition of 0.5 rounds up to 1.0?  If so the
// quantiser's (trunc(2u) + 1) >> 1 could be one instruction.  Prints the number of mismatches and the first few.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
__global__ void probe(uint32_t first, uint32_t count, unsigned long long *bad, uint32_t *ex) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const uint32_t bits = first + i;
    const float u = __builtin_bit_cast(float, bits);
    int r;
    asm volatile("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(u));
    const float t = __builtin_truncf(u);
    const int want = (int)(t + ((u - t) >= 0.5f ? 1.0f : 0.0f));  // f32::round for u >= 0 (kernels_image.hip quantise)
    if (r != want) {
        const unsigned long long k = atomicAdd(bad, 1ull);
        if (k < 8) {
            ex[2 * k] = bits;
            ex[2 * k + 1] = (uint32_t)r;
        }
    }
}
int main() {
    unsigned long long *bad, hb = 0;
    uint32_t *ex, hex[16] = {0};
    hipMalloc(&bad, 8);
    hipMalloc(&ex, 64);
    hipMemset(bad, 0, 8);
    hipMemset(ex, 0, 64);
    const uint32_t last = 0x48800000u;  // 2^18
    for (uint32_t first = 0; first < last; first += (1u << 28)) {
        const uint32_t n = last - first < (1u << 28) ? last - first : (1u << 28);
        hipLaunchKernelGGL(probe, dim3((n + 255) / 256), dim3(256), 0, 0, first, n, bad, ex);
    }
    // (-0.5, -0]: the clamp of quantise_regular leaves u > -0.5; all of these must give 0 (the same `want`: trunc = -0, frac < 0.5)
    hipLaunchKernelGGL(probe, dim3((0x3f000000u + 255) / 256), dim3(256), 0, 0, 0x80000000u, 0x3f000000u, bad, ex);
    hipDeviceSynchronize();
    hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
    hipMemcpy(hex, ex, 64, hipMemcpyDeviceToHost);
    printf("v_cvt_rpi_i32_f32 vs round-half-away on [0, 2^18) and (-0.5, -0]: %llu mismatches\n", hb);
    for (int k = 0; k < 8 && k < (int)hb; k++) printf("  bits %08x (%.9g): rpi %d\n", hex[2 * k], __builtin_bit_cast(float, hex[2 * k]), (int)hex[2 * k + 1]);
    return 0;
}
