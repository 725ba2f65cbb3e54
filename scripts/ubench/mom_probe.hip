// mom_probe.hip — measuring device / unit check for the moment-form mel epilogue (mel_moments_global, stft_wave.h) outside the FFT
// kernel: one wave per "frame", the amplitude row in LDS, the table from build_mel_moments.  Checks the device result against the
// same lane functions run on the host, and times N frames per wave.   usage: mom_probe [sr n_fft n_mel frames_per_wave]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../thesia_amd/csrc/host_math.cpp"
#include "../../thesia_amd/csrc/mel_fuse.h"
#include "../../thesia_amd/csrc/stft_wave.h"

using namespace th;

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(2);                                                      \
        }                                                                      \
    } while (0)

constexpr int WAVES = 8;
__global__ __launch_bounds__(64 * WAVES) void probe(const float *__restrict__ amp_g, uint32_t n_freq, uint32_t slab_f_len, const uint32_t *__restrict__ tab,
                                                     uint32_t n_groups, uint32_t n_mel, float *__restrict__ out, uint32_t pitch, uint32_t frames) {
    extern __shared__ float lds[];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63u;
    float *slab = lds + (size_t)wave * slab_f_len;
    for (uint32_t i = lane; i < slab_f_len; i += 64) slab[i] = i < n_freq ? amp_g[i] : __builtin_nanf("");
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    float mn = __builtin_inff();
    for (uint32_t f = 0; f < frames; f++) {
        const gptr<float> row = as_global(out) + ((size_t)(blockIdx.x * WAVES + wave) * frames + f) * pitch;
        mel_moments_global(lane, slab, as_global(tab), n_groups, [&](uint32_t m, float v) {
            if (m < n_mel) {
                row[m] = v;
                mn = fminf(mn, v);
            }
        });
    }
    if (mn == 12345.0f) out[0] = mn;
}

int main(int argc, char **argv) {
    const uint32_t sr = argc > 1 ? (uint32_t)std::atoi(argv[1]) : 96000, n_fft = argc > 2 ? (uint32_t)std::atoi(argv[2]) : 4096;
    uint32_t n_mel = argc > 3 ? (uint32_t)std::atoi(argv[3]) : 0;
    const uint32_t frames = argc > 4 ? (uint32_t)std::atoi(argv[4]) : 64;
    if (!n_mel) n_mel = (uint32_t)mel_default_n_mel(sr, n_fft);
    const uint32_t n_freq = n_fft / 2 + 1, nc = n_fft / 2, slab_len = 2 * (nc + nc / 16);
    const std::vector<float> fb = calc_mel_fb(sr, n_fft, n_mel, 0.f, -1.f, true);
    std::vector<float> lin, mf;
    mel_fb_points(sr, n_fft, n_mel, 0.f, -1.f, lin, mf);
    const MelMomHost h = build_mel_moments(fb.data(), lin.data(), mf.data(), n_freq, n_mel, slab_len, true);
    if (!h.ok) {
        std::printf("no moment form\n");
        return 1;
    }
    std::printf("sr %u n_fft %u n_mel %u: groups %u (W %u) taps %u max_dev %.3g max_amp %.3g words %zu\n", sr, n_fft, n_mel, h.n_groups, h.w_groups, h.taps,
                h.max_dev, h.max_amp, h.words.size());
    std::vector<float> amp(slab_len, NAN);
    srand(1);
    for (uint32_t k = 0; k < n_freq; k++) amp[k] = (float)(rand() % 10000) / 10000.f;
    // host: the lane functions
    std::vector<float> want(n_mel, NAN);
    {
        float carry = 0.f;
        for (uint32_t g = h.n_groups; g-- != 0;) {
            const uint32_t nw = h.words[MEL_MOM_HDR0 + 2 * g], n = nw & 0xffffu, off = h.words[MEL_MOM_HDR0 + 1 + 2 * g];
            const uint64_t *masks = reinterpret_cast<const uint64_t *>(h.words.data() + off + 256);
            MelMomLane s[64];
            for (uint32_t l = 0; l < 64; l++) {
                float prm[3], w1[2];
                std::memcpy(prm, &h.words[off + 4 * l + 1], 12);
                std::memcpy(w1, &h.words[off + 256 + 2 * l], 8);
                s[l] = (nw & MEL_MOM_FORM_W) ? mel_mom_w_lane(amp.data(), h.words[off + 4 * l], n, prm[0], prm[1], w1[0], w1[1])
                                             : mel_mom_lane_any(l, amp.data(), h.words[off + 4 * l], prm[0], prm[1], masks, n);
            }
            for (uint32_t l = 0; l < 64; l++) {
                float inv_d;
                std::memcpy(&inv_d, &h.words[off + 4 * l + 3], 4);
                if (64 * g + l < n_mel) want[64 * g + l] = mel_mom_combine(inv_d, s[l].R, l == 63 ? carry : s[l + 1].F);
            }
            carry = s[0].F;
        }
    }
    int dev = 0, n_cu = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev));
    const uint32_t pitch = (n_mel + 31) / 32 * 32, grid = (uint32_t)n_cu;
    float *d_amp, *d_out;
    uint32_t *d_tab;
    const size_t out_n = (size_t)grid * WAVES * frames * pitch;
    CK(hipMalloc(&d_amp, n_freq * 4));
    CK(hipMalloc(&d_out, out_n * 4));
    CK(hipMalloc(&d_tab, h.words.size() * 4));
    CK(hipMemcpy(d_amp, amp.data(), n_freq * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_tab, h.words.data(), h.words.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(d_out, 0xff, out_n * 4));
    const size_t lds = (size_t)WAVES * slab_len * 4;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int it = 0; it < 5; it++) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(probe, dim3(grid), dim3(64 * WAVES), lds, 0, d_amp, n_freq, slab_len, d_tab, h.n_groups, n_mel, d_out, pitch, frames);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::fmin(best, ms);
    }
    std::vector<float> got(out_n);
    CK(hipMemcpy(got.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    size_t bad = 0;
    for (size_t r = 0; r < (size_t)grid * WAVES * frames; r += 97)
        for (uint32_t m = 0; m < n_mel; m++) {
            const float a = got[r * pitch + m], b = want[m];
            if (!(a == b)) {
                if (bad < 8) std::printf("  row %zu mel %u: got %g want %g\n", r, m, a, b);
                bad++;
            }
            worst = std::fmax(worst, std::fabs((double)a - b));
        }
    const double nframes = (double)grid * WAVES * frames;
    std::printf("device == host lane functions: %s (mismatches %zu, worst abs %.3g); %.3f ms for %.0f frames = %.3f us per frame per wave (%.0f ns per frame chip-wide)\n",
                bad ? "NO" : "yes", bad, worst, best, nframes, best * 1e3 / frames, best * 1e6 / nframes);
    return bad ? 1 : 0;
}
