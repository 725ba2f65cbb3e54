// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this repo's
// kernels use (MI355X_MICROARCH.md §HBM: FETCH_SIZE is only calibrated for 16-B-per-lane reads).
// Each kernel streams a known number of bytes (2 GiB, far above the 256 MiB Infinity Cache).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <typename T>
__global__ void read_k(const T *__restrict__ in, float *out, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float acc = 0.f;
    for (; i < n; i += stride) {
        T v = in[i];
        acc += reinterpret_cast<const float *>(&v)[0];
    }
    if (acc == 12345.678f) out[0] = acc;
}
template <typename T>
__global__ void write_k(T *__restrict__ o, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    T v;
    for (int k = 0; k < (int)(sizeof(T) / 4); k++) reinterpret_cast<float *>(&v)[k] = (float)threadIdx.x;
    for (; i < n; i += stride) o[i] = v;
}
int main() {
    const size_t bytes = 2ull << 30;
    void *buf;
    float *out;
    (void)hipMalloc(&buf, bytes);
    (void)hipMalloc(&out, 4);
    (void)hipMemset(buf, 0, bytes);
    hipLaunchKernelGGL(read_k<float>, dim3(4096), dim3(256), 0, 0, (const float *)buf, out, bytes / 4);
    hipLaunchKernelGGL(read_k<float2>, dim3(4096), dim3(256), 0, 0, (const float2 *)buf, out, bytes / 8);
    hipLaunchKernelGGL(read_k<float4>, dim3(4096), dim3(256), 0, 0, (const float4 *)buf, out, bytes / 16);
    hipLaunchKernelGGL(write_k<float>, dim3(4096), dim3(256), 0, 0, (float *)buf, bytes / 4);
    hipLaunchKernelGGL(write_k<float2>, dim3(4096), dim3(256), 0, 0, (float2 *)buf, bytes / 8);
    hipLaunchKernelGGL(write_k<float4>, dim3(4096), dim3(256), 0, 0, (float4 *)buf, bytes / 16);
    (void)hipDeviceSynchronize();
    printf("streamed %zu bytes per kernel\n", bytes);
    return 0;
}
