// Probe (development tool): round trip of the plane exchange helpers of stft_wave.h on the GPU (12 waves, slabs above
// 24 KB like the kernel): st_group x4 -> read1 and st_group x4 (pitch 64) -> read2_paired, checked on the host.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../thesia_amd/csrc/stft_wave.h"
using namespace th;
using W = WaveFft<10>;
__global__ __launch_bounds__(768) void probe(float *out1, float *out2, int reps) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    cf32 *slab = reinterpret_cast<cf32 *>(smem + 24448) + (size_t)wave * W::SLAB_LEN;
    float *sf = reinterpret_cast<float *>(slab);
    for (int rep = 0; rep < reps; rep++) {
        cf32 v[16], z[16];
        for (int i = 0; i < 16; i++) v[i] = {(float)(wave * 100000 + lane * 100 + i + rep), -(float)(wave * 100000 + lane * 100 + i + rep)};
        W::st_group<W::PITCH1, 0>(sf, lane, v);
        W::st_group<W::PITCH1, 1>(sf, lane, v);
        W::st_group<W::PITCH1, 2>(sf, lane, v);
        W::st_group<W::PITCH1, 3>(sf, lane, v);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        W::read1(lane, z, slab);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (rep == reps - 1)
            for (int i = 0; i < 16; i++) {
                out1[((wave * 64 + lane) * 16 + i) * 2] = z[i].re;
                out1[((wave * 64 + lane) * 16 + i) * 2 + 1] = z[i].im;
            }
        W::st_group<W::PITCH2, 0>(sf, lane, v);
        W::st_group<W::PITCH2, 1>(sf, lane, v);
        W::st_group<W::PITCH2, 2>(sf, lane, v);
        W::st_group<W::PITCH2, 3>(sf, lane, v);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        cf32 za[2][4], zb[2][4];
        W::read2_paired(lane, za, zb, slab);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (rep == reps - 1)
            for (int q = 0; q < 2; q++)
                for (int r = 0; r < 4; r++) {
                    out2[(((wave * 64 + lane) * 2 + q) * 4 + r) * 4 + 0] = za[q][r].re;
                    out2[(((wave * 64 + lane) * 2 + q) * 4 + r) * 4 + 1] = za[q][r].im;
                    out2[(((wave * 64 + lane) * 2 + q) * 4 + r) * 4 + 2] = zb[q][r].re;
                    out2[(((wave * 64 + lane) * 2 + q) * 4 + r) * 4 + 3] = zb[q][r].im;
                }
    }
}
int main() {
    const int WV = 12, reps = 3;
    float *d1, *d2;
    hipMalloc(&d1, WV * 64 * 16 * 2 * 4);
    hipMalloc(&d2, WV * 64 * 2 * 4 * 4 * 4);
    const size_t lds = 24448 + (size_t)WV * W::SLAB_LEN * 8;
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64 * WV), lds, 0, d1, d2, reps);
    std::vector<float> h1(WV * 64 * 16 * 2), h2(WV * 64 * 2 * 4 * 4);
    hipMemcpy(h1.data(), d1, h1.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(h2.data(), d2, h2.size() * 4, hipMemcpyDeviceToHost);
    // plane k holds v[slot] with: group G = k & 3, i = k >> 2, slot = 4 G + i
    auto tag = [&](int wave, int lane, int k) { return (float)(wave * 100000 + lane * 100 + (4 * (k & 3) + (k >> 2)) + reps - 1); };
    int bad1 = 0, bad2 = 0;
    for (int w = 0; w < WV; w++)
        for (int l = 0; l < 64; l++) {
            const int c = l >> 2, a = l & 3;
            for (int r = 0; r < 16; r++) {
                const float want = tag(w, 16 * a + r, c);
                const float gr = h1[((w * 64 + l) * 16 + r) * 2], gi = h1[((w * 64 + l) * 16 + r) * 2 + 1];
                if (gr != want || gi != -want) {
                    if (bad1 < 10) printf("ex1 wave %d lane %d r %d: got (%g, %g) want %g\n", w, l, r, gr, gi, want);
                    bad1++;
                }
            }
            for (int q = 0; q < 2; q++) {
                const uint32_t ja = W::jj_a(l, q), jb = W::jj_b(l, q);
                for (int r = 0; r < 4; r++) {
                    const float wa = tag(w, 4 * (ja & 15) + r, ja >> 4), wb = tag(w, 4 * (jb & 15) + r, jb >> 4);
                    const float *g = &h2[(((w * 64 + l) * 2 + q) * 4 + r) * 4];
                    if (g[0] != wa || g[1] != -wa || g[2] != wb || g[3] != -wb) {
                        if (bad2 < 10) printf("ex2 wave %d lane %d q %d r %d: got (%g, %g | %g, %g) want %g | %g\n", w, l, q, r, g[0], g[1], g[2], g[3], wa, wb);
                        bad2++;
                    }
                }
            }
        }
    printf("planes probe: %d bad of %d (exchange 1), %d bad of %d (exchange 2)\n", bad1, WV * 64 * 16, bad2, WV * 64 * 8);
    return 0;
}
