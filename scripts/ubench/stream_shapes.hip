// Micro-benchmark (development tool, round 3): which ACCESS STRUCTURE does the memory system serve fastest for the STFT
// kernel's traffic — 128 channels x 1.44 M samples read once (0.74 GB), 360064 rows of 1025 floats written at pitch 1056
// (1.52 GB) — with no arithmetic at all?
//   A  "row streams"   (what stft_wave_kernel does): 3072 persistent waves, each walks chunks of 30 consecutive frames of one
//                      channel: per frame one hop (2 KB) loaded a frame ahead, one row (17 dword stores) written.
//   B  "sweep"         a workgroup of W waves takes the next GROUP of W x R consecutive frames from an in-order queue, loads
//                      the group's audio span cooperatively (16 B per lane, linear), then wave w writes rows w R .. w R + R - 1.
//                      Everything in flight chip-wide is one compact window that sweeps linearly through input and output.
//   C  stores alone / loads alone for both structures; plain linear fill and read for reference.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_shapes stream_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <vector>
#include <string>
#include <chrono>
#include <cstdlib>

constexpr uint32_t N_CH = 128, N_SAMP = 1440000, HOP = 512, NFFT = 2048, T = 2813, PITCH = 1056, H = 1025;
constexpr uint32_t FA = 2, FB = T - 2;  // interior frames [FA, FB): the span [f hop - 1024, + 2048) lies inside the channel

typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

__device__ inline void store_row(float *row, uint32_t lane, float v, int mode) {
    if (mode == 0) {  // 16 aligned dword stores + completion of the last line
#pragma unroll
        for (int j = 0; j < 16; j++) row[lane + 64 * j] = v;
        if (lane < 32) row[1024 + lane] = v;
    } else {  // 16 bytes per lane
#pragma unroll
        for (int j = 0; j < 4; j++) reinterpret_cast<f4 *>(row)[lane + 64 * j] = f4{v, v, v, v};
        if (lane < 8) reinterpret_cast<f4 *>(row)[256 + lane] = f4{v, v, v, v};
    }
}

// ---- A: row streams
template <int WAVES, int DO_LOAD, int DO_STORE, int SMODE>
__global__ __launch_bounds__(64 * WAVES) void k_rows(const float *__restrict__ wav, float *__restrict__ out, uint32_t chunk,
                                                     uint32_t chunks_per_ch, uint32_t *queue, float *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = chunks_per_ch * N_CH;
    float acc = 0.f;
    uint32_t c = blockIdx.x * WAVES + (threadIdx.x >> 6);
    const uint32_t n_waves = gridDim.x * WAVES;
    while (c < n_chunks) {
        const uint32_t ch = c / chunks_per_ch, f0 = FA + (c % chunks_per_ch) * chunk;
        const uint32_t f1 = min(FB, f0 + chunk);
        const float *x = wav + (size_t)ch * N_SAMP;
        float *o = out + (size_t)ch * T * PITCH;
        f2 nxt[4];
        if (DO_LOAD) {  // first frame in full: 16 loads
#pragma unroll
            for (int m = 0; m < 16; m++) {
                f2 v = *reinterpret_cast<const f2 *>(x + (size_t)f0 * HOP - 1024 + 2 * (lane + 64 * m));
                acc += v.x + v.y;
            }
        }
        for (uint32_t f = f0; f < f1; f++) {
            if (DO_LOAD) {
                const uint32_t fn = f + 1 < f1 ? f + 1 : f;
#pragma unroll
                for (int m = 0; m < 4; m++) nxt[m] = *reinterpret_cast<const f2 *>(x + (size_t)fn * HOP + 512 + 2 * (lane + 64 * m));
            }
            if (DO_STORE) store_row(o + (size_t)f * PITCH, lane, acc, SMODE);
            if (DO_LOAD) {
#pragma unroll
                for (int m = 0; m < 4; m++) acc += nxt[m].x + nxt[m].y;
            }
        }
        uint32_t nc = 0;
        if (lane == 0) nc = atomicAdd(queue, 1u);
        c = n_waves + __builtin_amdgcn_readfirstlane(nc);
    }
    if (acc == 123.456f) *sink = acc;
}

// ---- A': row streams with static round-robin chunks of R frames: wave w takes chunks w, w + n_waves, ... (compact window)
template <int WAVES, int R, int SMODE>
__global__ __launch_bounds__(64 * WAVES) void k_rows_rr(const float *__restrict__ wav, float *__restrict__ out, uint32_t chunks_per_ch, float *sink) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n_chunks = chunks_per_ch * N_CH, n_waves = gridDim.x * WAVES;
    float acc = 0.f;
    for (uint32_t c = blockIdx.x * WAVES + (threadIdx.x >> 6); c < n_chunks; c += n_waves) {
        const uint32_t ch = c / chunks_per_ch, f0 = FA + (c % chunks_per_ch) * R;
        const float *x = wav + (size_t)ch * N_SAMP + (size_t)f0 * HOP - 1024;
        float *o = out + (size_t)ch * T * PITCH;
        const float *end = wav + (size_t)N_SAMP * N_CH - 2;
#pragma unroll
        for (int m = 0; m < 16 + 4 * (R - 1); m++) {
            const float *p = x + 2 * (lane + 64 * m);
            p = p < end ? p : end;
            f2 v = *reinterpret_cast<const f2 *>(p);
            acc += v.x + v.y;
        }
#pragma unroll
        for (int r = 0; r < R; r++)
            if (f0 + r < FB) store_row(o + (size_t)(f0 + r) * PITCH, lane, acc, SMODE);
    }
    if (acc == 123.456f) *sink = acc;
}

// ---- B: sweep.  Group = WAVES * R consecutive frames of one channel; span = (G - 1) hop + n_fft samples
template <int WAVES, int R, int DO_LOAD, int DO_STORE, int SMODE, int SYNC, int PERM = 0, int OWN = 0>
__global__ __launch_bounds__(64 * WAVES) void k_sweep(const float *__restrict__ wav, float *__restrict__ out, uint32_t groups_per_ch,
                                                      uint32_t *queue, float *sink, uint32_t perm_mul = 1) {
    constexpr uint32_t G = WAVES * R, SPAN = (G - 1) * HOP + NFFT, NT = 64 * WAVES;
    constexpr uint32_t NLD = (SPAN / 4 + NT - 1) / NT;
    __shared__ uint32_t g_s;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t n_groups = groups_per_ch * N_CH;
    float acc = 0.f;
    uint32_t g = blockIdx.x;
    while (g < n_groups) {
        const uint32_t gp = PERM ? (uint32_t)(((uint64_t)g * perm_mul) % n_groups) : g;  // PERM: groups in scattered order
        const uint32_t ch = gp / groups_per_ch, f0 = FA + (gp % groups_per_ch) * G;
        const float *x = wav + (size_t)ch * N_SAMP + (size_t)f0 * HOP - 1024;
        float *o = out + (size_t)ch * T * PITCH;
        if (DO_LOAD && OWN) {  // every wave loads the span of its own R frames: 8-byte loads, overlap served by L1 / L2
            const float *xw = x + (size_t)wave * R * HOP;
            const float *end = wav + (size_t)N_SAMP * N_CH - 2;
#pragma unroll
            for (int m = 0; m < 16 + 4 * (R - 1); m++) {
                const float *p = xw + 2 * (lane + 64 * m);
                p = p < end ? p : end;
                f2 v = *reinterpret_cast<const f2 *>(p);
                acc += v.x + v.y;
            }
        } else if (DO_LOAD) {
            f4 v[NLD];
#pragma unroll
            for (uint32_t m = 0; m < NLD; m++) {
                uint32_t i = threadIdx.x + NT * m;
                i = i < SPAN / 4 ? i : SPAN / 4 - 1;
                const size_t lim = (size_t)N_SAMP * N_CH / 4 - 1;
                size_t idx = ((size_t)(x - wav)) / 4 + i;  // (x is 16-byte aligned: f0 * 512 - 1024)
                idx = idx < lim ? idx : lim;
                v[m] = reinterpret_cast<const f4 *>(wav)[idx];
            }
#pragma unroll
            for (uint32_t m = 0; m < NLD; m++) acc += v[m].x + v[m].y + v[m].z + v[m].w;
        }
        if (SYNC) __syncthreads();
        if (DO_STORE) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t f = f0 + wave * R + r;
                if (f < FB) store_row(o + (size_t)f * PITCH, lane, acc, SMODE);
            }
        }
        if (threadIdx.x == 0) g_s = gridDim.x + atomicAdd(queue, 1u);
        __syncthreads();
        g = g_s;
        if (SYNC) __syncthreads();
    }
    if (acc == 123.456f) *sink = acc;
}

// ---- reference: linear fill / read / copy-like 1:2 mix
__global__ void k_fill(f4 *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = f4{1.f, 2.f, 3.f, 4.f};
}
__global__ void k_read(const f4 *in, size_t n, float *sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = in[i];
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 123.456f) *sink = acc;
}
__global__ void k_mix(const f4 *in, f4 *out, size_t n_in, float *sink) {  // read n_in, write 2 n_in, all linear
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (size_t)gridDim.x * blockDim.x) {
        f4 v = in[i];
        out[2 * i] = v;
        out[2 * i + 1] = v;
    }
}

template <typename F>
static float time_it(F &&launch, uint32_t *queue) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<float> ms;
    for (int i = 0; i < 12; i++) {
        hipMemsetAsync(queue, 0, 4);
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float t;
        hipEventElapsedTime(&t, e0, e1);
        if (i >= 2) ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    return ms[ms.size() / 2];
}

int main(int argc, char **argv) {
    float *wav, *out, *sink;
    uint32_t *queue;
    const size_t wav_b = (size_t)N_CH * N_SAMP * 4, out_b = (size_t)N_CH * T * PITCH * 4;
    hipMalloc(&wav, wav_b);
    hipMalloc(&out, out_b);
    hipMalloc(&sink, 4);
    hipMalloc(&queue, 4);
    hipMemset(wav, 0, wav_b);
    hipMemset(out, 0, out_b);
    const double rd = (double)N_CH * (FB - FA) * HOP * 4, wr = (double)N_CH * (FB - FA) * PITCH * 4;
    if (argc >= 3 && std::string(argv[1]) == "loop") {  // keep one variant running (power_probe.sh reads the package power next to it)
        const std::string v = argv[2];
        const double secs = argc >= 4 ? atof(argv[3]) : 14.0;
        const auto t0 = std::chrono::steady_clock::now();
        long it = 0;
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        double ms_sum = 0;
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
            hipEventRecord(e0);
            for (int k = 0; k < 50; k++) {
                hipMemsetAsync(queue, 0, 4);
                if (v == "A") hipLaunchKernelGGL((k_rows<12, 1, 1, 0>), dim3(256), dim3(768), 0, 0, wav, out, 30u, (FB - FA + 29) / 30, queue, sink);
                else if (v == "B12x4") hipLaunchKernelGGL((k_sweep<12, 4, 1, 1, 0, 1, 0, 1>), dim3(256), dim3(768), 0, 0, wav, out, (FB - FA + 47) / 48, queue, sink, 1u);
                else if (v == "B8x4") hipLaunchKernelGGL((k_sweep<8, 4, 1, 1, 0, 1, 0, 1>), dim3(256), dim3(512), 0, 0, wav, out, (FB - FA + 31) / 32, queue, sink, 1u);
                else if (v == "B16x1") hipLaunchKernelGGL((k_sweep<16, 1, 1, 1, 0, 1>), dim3(256), dim3(1024), 0, 0, wav, out, (FB - FA + 15) / 16, queue, sink, 1u);
                else return 2;
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float t;
            hipEventElapsedTime(&t, e0, e1);
            ms_sum += t;
            it += 50;
        }
        printf("%s: %ld launches, %.3f ms each, %.0f GB/s\n", v.c_str(), it, ms_sum / it, (rd + wr) / (ms_sum / it) / 1e6);
        return 0;
    }
    auto rep = [&](const char *name, float ms, double bytes) { printf("%-78s %.3f ms  %5.0f GB/s\n", name, ms, bytes / ms / 1e6); fflush(stdout); };
    rep("linear fill 16 B/lane (1.52 GB)", time_it([&] { hipLaunchKernelGGL(k_fill, dim3(256 * 8), dim3(512), 0, 0, (f4 *)out, out_b / 16); }, queue), (double)out_b);
    rep("linear read 16 B/lane (0.74 GB)", time_it([&] { hipLaunchKernelGGL(k_read, dim3(256 * 8), dim3(512), 0, 0, (const f4 *)wav, wav_b / 16, sink); }, queue), (double)wav_b);
    rep("linear read 0.74 GB + write 1.47 GB", time_it([&] { hipLaunchKernelGGL(k_mix, dim3(256 * 8), dim3(512), 0, 0, (const f4 *)wav, (f4 *)out, wav_b / 16, sink); }, queue), 3.0 * wav_b);
#define ROWS(W, L, S, M, chunk, label)                                                                                      \
    {                                                                                                                       \
        const uint32_t cpc = (FB - FA + chunk - 1) / chunk;                                                                 \
        rep(label, time_it([&] { hipLaunchKernelGGL((k_rows<W, L, S, M>), dim3(256), dim3(64 * W), 0, 0, wav, out, chunk, cpc, queue, sink); }, queue), \
            (L ? rd : 0) + (S ? wr : 0));                                                                                   \
    }
    ROWS(12, 1, 1, 0, 30, "A rows: 12 waves, chunk 30, load + dword stores")
    ROWS(12, 1, 1, 1, 30, "A rows: 12 waves, chunk 30, load + 16-byte stores")
    ROWS(12, 0, 1, 0, 30, "A rows: 12 waves, stores alone (dword)")
    ROWS(12, 0, 1, 1, 30, "A rows: 12 waves, stores alone (16-byte)")
    ROWS(12, 1, 0, 0, 30, "A rows: 12 waves, loads alone")
    ROWS(8, 1, 1, 0, 30, "A rows:  8 waves, chunk 30, load + dword stores")
    ROWS(16, 1, 1, 0, 30, "A rows: 16 waves, chunk 30, load + dword stores")
#define SWEEP(W, R, L, S, M, Y, label)                                                                                      \
    {                                                                                                                       \
        const uint32_t gpc = (FB - FA + W * R - 1) / (W * R);                                                               \
        rep(label, time_it([&] { hipLaunchKernelGGL((k_sweep<W, R, L, S, M, Y>), dim3(256), dim3(64 * W), 0, 0, wav, out, gpc, queue, sink); }, queue), \
            (L ? rd : 0) + (S ? wr : 0));                                                                                   \
    }
    SWEEP(12, 1, 1, 1, 0, 1, "B sweep: 12 waves x 1 row, load + dword stores, barriers")
    SWEEP(12, 1, 1, 1, 1, 1, "B sweep: 12 waves x 1 row, load + 16-byte stores, barriers")
    SWEEP(12, 2, 1, 1, 0, 1, "B sweep: 12 waves x 2 rows, load + dword stores, barriers")
    SWEEP(12, 2, 1, 1, 1, 1, "B sweep: 12 waves x 2 rows, load + 16-byte stores, barriers")
    SWEEP(12, 4, 1, 1, 0, 1, "B sweep: 12 waves x 4 rows, load + dword stores, barriers")
    SWEEP(16, 1, 1, 1, 0, 1, "B sweep: 16 waves x 1 row, load + dword stores, barriers")
    SWEEP(16, 2, 1, 1, 1, 1, "B sweep: 16 waves x 2 rows, load + 16-byte stores, barriers")
    SWEEP(8, 2, 1, 1, 0, 1, "B sweep:  8 waves x 2 rows, load + dword stores, barriers")
    SWEEP(12, 2, 0, 1, 0, 1, "B sweep: 12 waves x 2 rows, stores alone (dword)")
    SWEEP(12, 2, 0, 1, 1, 1, "B sweep: 12 waves x 2 rows, stores alone (16-byte)")
    SWEEP(12, 2, 1, 0, 0, 1, "B sweep: 12 waves x 2 rows, loads alone")
    SWEEP(16, 1, 1, 1, 0, 0, "B sweep: 16 waves x 1 row, load + dword stores, NO barriers")
    SWEEP(12, 2, 1, 1, 0, 0, "B sweep: 12 waves x 2 rows, load + dword stores, NO barriers")
#define SWEEPX(W, R, PERM, OWN, label)                                                                                      \
    {                                                                                                                       \
        const uint32_t gpc = (FB - FA + W * R - 1) / (W * R), ng = gpc * N_CH;                                              \
        uint32_t mul = (uint32_t)(ng * 0.6180339887);                                                                       \
        auto gcd = [](uint32_t a, uint32_t b) { while (b) { uint32_t t = a % b; a = b; b = t; } return a; };               \
        while (gcd(mul, ng) != 1) mul++;                                                                                    \
        rep(label, time_it([&] { hipLaunchKernelGGL((k_sweep<W, R, 1, 1, 0, 1, PERM, OWN>), dim3(256), dim3(64 * W), 0, 0, wav, out, gpc, queue, sink, mul); }, queue), rd + wr); \
    }
    SWEEPX(16, 1, 1, 0, "B sweep: 16 waves x 1 row, groups in SCATTERED order")
    SWEEPX(12, 2, 1, 0, "B sweep: 12 waves x 2 rows, groups in SCATTERED order")
    SWEEPX(12, 4, 1, 0, "B sweep: 12 waves x 4 rows, groups in SCATTERED order")
    SWEEPX(16, 1, 0, 1, "B sweep: 16 waves x 1 row, every wave loads its OWN frame (8 KB)")
    SWEEPX(12, 2, 0, 1, "B sweep: 12 waves x 2 rows, every wave loads its OWN span (10 KB)")
    SWEEPX(12, 4, 0, 1, "B sweep: 12 waves x 4 rows, every wave loads its OWN span (14 KB)")
    SWEEPX(12, 2, 1, 1, "B sweep: 12 waves x 2 rows, OWN span, SCATTERED order")
    SWEEPX(12, 8, 0, 1, "B sweep: 12 waves x 8 rows, OWN span")
    SWEEPX(12, 16, 0, 1, "B sweep: 12 waves x 16 rows, OWN span")
    SWEEPX(12, 30, 0, 1, "B sweep: 12 waves x 30 rows, OWN span")
    SWEEPX(8, 4, 0, 1, "B sweep:  8 waves x 4 rows, OWN span")
    SWEEPX(16, 4, 0, 1, "B sweep: 16 waves x 4 rows, OWN span")
    {
        const uint32_t gpc = (FB - FA + 6 * 4 - 1) / (6 * 4);
        rep("B sweep: 2 x 6 waves per CU x 4 rows, OWN span", time_it([&] { hipLaunchKernelGGL((k_sweep<6, 4, 1, 1, 0, 1, 0, 1>), dim3(512), dim3(64 * 6), 0, 0, wav, out, gpc, queue, sink, 1u); }, queue), rd + wr);
        const uint32_t gpc8 = (FB - FA + 6 * 8 - 1) / (6 * 8);
        rep("B sweep: 2 x 6 waves per CU x 8 rows, OWN span", time_it([&] { hipLaunchKernelGGL((k_sweep<6, 8, 1, 1, 0, 1, 0, 1>), dim3(512), dim3(64 * 6), 0, 0, wav, out, gpc8, queue, sink, 1u); }, queue), rd + wr);
    }
#define RR(W, R, label)                                                                                                     \
    {                                                                                                                       \
        const uint32_t cpc = (FB - FA + R - 1) / R;                                                                         \
        rep(label, time_it([&] { hipLaunchKernelGGL((k_rows_rr<W, R, 0>), dim3(256), dim3(64 * W), 0, 0, wav, out, cpc, sink); }, queue), rd + wr); \
    }
    RR(12, 1, "A' static round-robin chunks of 1 frame, own loads (8 KB per frame)")
    RR(12, 2, "A' static round-robin chunks of 2 frames, own loads")
    RR(12, 4, "A' static round-robin chunks of 4 frames, own loads")
    RR(12, 8, "A' static round-robin chunks of 8 frames, own loads")
    RR(16, 2, "A' static round-robin chunks of 2 frames, own loads, 16 waves")
    {   // two workgroups per CU (6 waves each)
        const uint32_t gpc = (FB - FA + 6 * 2 - 1) / (6 * 2);
        rep("B sweep: 2 x 6 waves per CU x 2 rows, load + dword stores", time_it([&] { hipLaunchKernelGGL((k_sweep<6, 2, 1, 1, 0, 1>), dim3(512), dim3(64 * 6), 0, 0, wav, out, gpc, queue, sink); }, queue), rd + wr);
        const uint32_t gpc4 = (FB - FA + 4 * 2 - 1) / (4 * 2);
        rep("B sweep: 3 x 4 waves per CU x 2 rows, load + dword stores", time_it([&] { hipLaunchKernelGGL((k_sweep<4, 2, 1, 1, 0, 1>), dim3(768), dim3(64 * 4), 0, 0, wav, out, gpc4, queue, sink); }, queue), rd + wr);
    }
    return 0;
}
