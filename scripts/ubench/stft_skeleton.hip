// Micro-benchmark (development tool): the memory skeleton of the wave STFT kernel — persistent waves, chunk queue,
// per frame: the input loads of one hop (prefetched a frame ahead), optional stand-in compute (dependent FMA rounds and
// LDS write/read rounds), the output row stores in several shapes.  No FFT: what is the floor the access pattern
// itself sets, and what does each load / store shape cost next to a given amount of VALU / LDS work?
//   LOADM  0 none | 1 four 8-byte loads per lane (lane + 64 m layout) | 2 two 16-byte loads per lane
//   STOREM 0 none | 1 kernel pattern (8 aligned + 8 mirrored dword stores + bin 512 + tail line) | 2 16 aligned dword
//          stores + tail | 3 four 16-byte stores + tail line
//   FMA    dependent-FMA rounds of 16 instructions per frame         LDSR  rounds of (8 ds_write_b128 + 16 ds_read_b64)
// build: hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o stft_skeleton stft_skeleton.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <algorithm>
#include <type_traits>
#include <ctime>
#include <vector>

using gf = __attribute__((address_space(1))) float;
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));
using gf2 = __attribute__((address_space(1))) v2f;
using gf4 = __attribute__((address_space(1))) v4f;

#if !defined(CHUNK_FRAMES)
#define CHUNK_FRAMES 32
#endif
// -DNYQ_PLANE (round 5, VERDICT r4 #5a): rows of exactly 4096 bytes (bins 0 .. 1023, no padding, no partial line) and the
// Nyquist bins of a channel in a plane of their own behind the rows — a wave keeps the bin of each frame of its chunk in one
// lane and stores them together at the end of the chunk
#if defined(NYQ_PLANE)
constexpr uint32_t N_FFT = 2048, HOP = 512, H = 1025, PITCH = 1024, CHUNK = CHUNK_FRAMES;
#else
constexpr uint32_t N_FFT = 2048, HOP = 512, H = 1025, PITCH = 1056, CHUNK = CHUNK_FRAMES;
#endif

template <int LOADM, int STOREM, int FMA, int LDSR, int WAVES, int LDSM = 0>
__global__ __launch_bounds__(64 * WAVES) void k(const float *wav_, float *spec_, uint32_t n_chan, uint32_t n_samples,
                                                uint32_t T, uint32_t chunks_per_chan, uint32_t *queue, float seed) {
    extern __shared__ v2f lds[];
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f *slab = lds + wave * 1100;
    const uint32_t n_chunks = n_chan * chunks_per_chan;
    bool first = true;
    uint32_t round_no = 0;
    (void)round_no;
    for (;;) {
        uint32_t c = blockIdx.x * WAVES + wave;
#if defined(STATIC_ROUNDS)
        // chunk t goes to wave t mod W in round t / W, no queue: waves that advance at the same pace write neighbouring rows
        // at the same time (DRAM locality of the store stream)
        c += round_no * gridDim.x * WAVES;
        round_no++;
#else
        if (!first) {
            if (lane == 0) c = atomicAdd(queue, 1u);
            c = __builtin_amdgcn_readfirstlane(c) + gridDim.x * WAVES;
        }
#endif
        first = false;
        if (c >= n_chunks) break;
        const uint32_t ch = c / chunks_per_chan, f0 = 2 + (c % chunks_per_chan) * CHUNK;  // interior frames only
        const uint32_t f1 = min(f0 + CHUNK, T - 2);
        const gf *wav = (const gf *)(wav_ + (size_t)ch * n_samples);
        gf *spec = (gf *)(spec_ + (size_t)ch * T * PITCH);
#if defined(NYQ_PLANE)
        gf *plane = (gf *)(spec_ + (size_t)n_chan * T * PITCH + (size_t)ch * T);
        float nyq = 0.f;
#endif
        v2f x[16];
        float acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) {
            x[i] = v2f{0.f, 0.f};
            acc[i] = seed + i;
        }
        {
            const int64_t e0 = (int64_t)f0 * HOP - N_FFT / 2;
            if (LOADM == 1 || LOADM == 3) {
                const uint32_t L = LOADM == 3 ? 4 * (lane & 15) + (lane >> 4) : lane;  // 3: 8-byte loads at a 32-byte lane stride
#pragma unroll
                for (int m = 0; m < 16; m++) x[m] = *(const gf2 *)(wav + e0 + 2 * (L + 64 * m));
            } else if (LOADM == 2) {
#pragma unroll
                for (int m = 0; m < 8; m++) {
                    const v4f v = *(const gf4 *)(wav + e0 + 4 * (lane + 64 * m));
                    x[2 * m] = v2f{v.x, v.y};
                    x[2 * m + 1] = v2f{v.z, v.w};
                }
            }
        }
#if defined(LOOKAHEAD2)
        // the new hop of frame f + 1 was requested TWO frames ago (buffers ya / yb alternate): gfx9 has one in-order
        // vmcnt for loads and stores, so a wait for loads issued one frame ago also waits for the stores of the frame
        // before that
        const uint32_t LL = LOADM == 3 ? 4 * (lane & 15) + (lane >> 4) : lane;
        v2f ya[4], yb[4];
        {
            const uint32_t fa = min(f0 + 1, f1 - 1), fb = min(f0 + 2, f1 - 1);
#pragma unroll
            for (int m = 0; m < 4; m++) ya[m] = *(const gf2 *)(wav + ((int64_t)fa * HOP - N_FFT / 2) + 2 * (LL + 64 * (12 + m)));
#pragma unroll
            for (int m = 0; m < 4; m++) yb[m] = *(const gf2 *)(wav + ((int64_t)fb * HOP - N_FFT / 2) + 2 * (LL + 64 * (12 + m)));
        }
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (uint32_t f = f0; f < f1; f++) {
            // consume the frame in registers: fold all 16 slots into the accumulators (stands for the window multiply)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = acc[i] * 0.999f + (x[i].x + x[i].y);
            // prefetch the next frame's new hop (slots 12..15 after a shift by 4)
            const uint32_t fn = f + 1 < f1 ? f + 1 : f;
            const int64_t e0n = (int64_t)fn * HOP - N_FFT / 2;
#pragma unroll
            for (int m = 0; m < 12; m++) x[m] = x[m + 4];
#if defined(LOOKAHEAD2)
            {
                const uint32_t f3 = min(f + 3, f1 - 1);
                const int64_t e3 = (int64_t)f3 * HOP - N_FFT / 2;
                if (((f - f0) & 1u) == 0) {
#pragma unroll
                    for (int m = 0; m < 4; m++) x[12 + m] = ya[m];
#pragma unroll
                    for (int m = 0; m < 4; m++) ya[m] = *(const gf2 *)(wav + e3 + 2 * (LL + 64 * (12 + m)));
                } else {
#pragma unroll
                    for (int m = 0; m < 4; m++) x[12 + m] = yb[m];
#pragma unroll
                    for (int m = 0; m < 4; m++) yb[m] = *(const gf2 *)(wav + e3 + 2 * (LL + 64 * (12 + m)));
                }
            }
            if (false) {
#else
            if (LOADM == 1 || LOADM == 3) {
#endif
                const uint32_t L = LOADM == 3 ? 4 * (lane & 15) + (lane >> 4) : lane;
#pragma unroll
                for (int m = 12; m < 16; m++) x[m] = *(const gf2 *)(wav + e0n + 2 * (L + 64 * m));
            } else if (LOADM == 2) {
#pragma unroll
                for (int m = 6; m < 8; m++) {
                    const v4f v = *(const gf4 *)(wav + e0n + 4 * (lane + 64 * m));
                    x[2 * m] = v2f{v.x, v.y};
                    x[2 * m + 1] = v2f{v.z, v.w};
                }
            }
            // stand-in compute
            if (FMA >= 0) {
#pragma unroll 1
                for (int r = 0; r < FMA; r++) {
#pragma unroll
                    for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(acc[i], 1.0001f, 0.25f);
                }
            } else {  // the same arithmetic as -FMA rounds, as 8 v_pk_fma_f32 per round
                v2f pa[8];
#pragma unroll
                for (int i = 0; i < 8; i++) pa[i] = v2f{acc[2 * i], acc[2 * i + 1]};
                const v2f m = v2f{1.0001f, 1.0001f}, c = v2f{0.25f, 0.25f};
#pragma unroll 1
                for (int r = 0; r < -FMA; r++) {
#pragma unroll
                    for (int i = 0; i < 8; i++) pa[i] = __builtin_elementwise_fma(pa[i], m, c);
                }
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    acc[2 * i] = pa[i].x;
                    acc[2 * i + 1] = pa[i].y;
                }
            }
#pragma unroll 1
            for (int r = 0; r < LDSR; r++) {
                if (LDSM == 0) {  // the kernel's exchange 1: 8 x ds_write_b128 (2-way bank conflicts), 16 x ds_read_b64
#pragma unroll
                    for (int i = 0; i < 8; i++)
                        *(v4f *)&slab[(lane * 16 + 2 * (lane >> 1)) + 2 * i] = v4f{acc[2 * i], acc[2 * i + 1], acc[2 * i], acc[2 * i + 1]};
                } else if (LDSM == 3) {  // 16 x ds_write_b64, pitch 17 slots per lane: conflict-free writes
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint64_t u = (uint64_t)__builtin_bit_cast(uint32_t, acc[i]) | ((uint64_t)__builtin_bit_cast(uint32_t, acc[15 - i]) << 32);
                        *(volatile __attribute__((address_space(3))) uint64_t *)(&slab[lane * 17 + i]) = u;
                    }
                } else {  // 32 x ds_write_addtid_b32 into 32 planes of 64 (+4) dwords
                    const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)slab);
#define WA(i, o) "ds_write_addtid_b32 %" #i " offset:" #o "\n\t"
                    asm volatile("s_mov_b32 m0, %16\n\t" WA(0, 0) WA(1, 272) WA(2, 544) WA(3, 816) WA(4, 1088) WA(5, 1360) WA(6, 1632) WA(7, 1904)
                                 WA(8, 2176) WA(9, 2448) WA(10, 2720) WA(11, 2992) WA(12, 3264) WA(13, 3536) WA(14, 3808) WA(15, 4080)
                                 :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]),
                                 "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]),
                                 "s"(base) : "memory");
                    asm volatile("s_mov_b32 m0, %16\n\t" WA(0, 4352) WA(1, 4624) WA(2, 4896) WA(3, 5168) WA(4, 5440) WA(5, 5712) WA(6, 5984) WA(7, 6256)
                                 WA(8, 6528) WA(9, 6800) WA(10, 7072) WA(11, 7344) WA(12, 7616) WA(13, 7888) WA(14, 8160) WA(15, 8432)
                                 :: "v"(acc[15]), "v"(acc[14]), "v"(acc[13]), "v"(acc[12]), "v"(acc[11]), "v"(acc[10]), "v"(acc[9]), "v"(acc[8]),
                                 "v"(acc[7]), "v"(acc[6]), "v"(acc[5]), "v"(acc[4]), "v"(acc[3]), "v"(acc[2]), "v"(acc[1]), "v"(acc[0]),
                                 "s"(base) : "memory");
#undef WA
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                if (LDSM == 0) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint64_t v = *(volatile __attribute__((address_space(3))) uint64_t *)(&slab[lane + 2 * (lane >> 5) + 68 * i]);
                        acc[i] += __builtin_bit_cast(float, (uint32_t)v);
                    }
                } else if (LDSM == 3) {
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const uint32_t idx = lane + 64 * i;
                        const uint64_t v = *(volatile __attribute__((address_space(3))) uint64_t *)(&slab[(idx >> 4) * 17 + (lane & 15)]);
                        acc[i] += __builtin_bit_cast(float, (uint32_t)v);
                    }
                } else if (LDSM == 1) {  // 32 x ds_read_b32: plane c = lane >> 2 (+ 16), element 4 r + (lane & 3)
                    const float *sf = reinterpret_cast<const float *>(slab);
#pragma unroll
                    for (int i = 0; i < 16; i++) {
                        const float v0 = *(volatile __attribute__((address_space(3))) float *)(&sf[(lane >> 2) * 68 + 4 * i + (lane & 3)]);
                        const float v1 = *(volatile __attribute__((address_space(3))) float *)(&sf[(16 + (lane >> 2)) * 68 + 4 * i + (lane & 3)]);
                        acc[i] += v0 + v1;
                    }
                } else {  // 8 x ds_read_b128: 4 consecutive dwords of 8 planes
                    const float *sf = reinterpret_cast<const float *>(slab);
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const v4f v = *(const v4f *)(&sf[(4 * i + (lane >> 4)) * 68 + 4 * (lane & 15)]);
                        acc[2 * i] += v.x + v.y;
                        acc[2 * i + 1] += v.z + v.w;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
            gf *row = spec + (size_t)f * PITCH;
            if (STOREM == 1) {
#pragma unroll
                for (int j = 0; j < 8; j++) row[lane + 64 * j] = acc[j];
#if defined(NYQ_PLANE)
                if (lane) row[1024 - lane] = acc[8];
#pragma unroll
                for (int j = 1; j < 8; j++) row[1024 - lane - 64 * j] = acc[8 + j];
                if (lane == 0) row[512] = acc[0];
                nyq = lane == f - f0 ? __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc[8])) * 1.0f : nyq;
#else
#pragma unroll
                for (int j = 0; j < 8; j++) row[1024 - lane - 64 * j] = acc[8 + j];
                if (lane == 0) row[512] = acc[0];
                if (lane - 1u < 31u) row[1024 + lane] = 0.f;
#endif
            } else if (STOREM == 4 || STOREM == 5) {
                // adjacent-bin pairs: A side 4 aligned 8-byte stores (bins 2 l + 256 s, + 1); B side the mirrored pairs
                // (bins 1023 - 2 l - 256 s, + 1): 4 misaligned 8-byte stores (mode 4) or 8 dword stores (mode 5)
                typedef v2f __attribute__((aligned(4))) v2f_u;
                using gf2u = __attribute__((address_space(1))) v2f_u;
#pragma unroll
                for (int j = 0; j < 4; j++) *(gf2 *)(row + 2 * lane + 256 * j) = v2f{acc[2 * j], acc[2 * j + 1]};
                if (STOREM == 4) {
#pragma unroll
                    for (int j = 0; j < 4; j++) *(gf2u *)(row + 1023 - 2 * lane - 256 * j) = v2f{acc[8 + 2 * j], acc[9 + 2 * j]};
                } else {
#pragma unroll
                    for (int j = 0; j < 4; j++) {
                        row[1023 - 2 * lane - 256 * j] = acc[8 + 2 * j];
                        row[1024 - 2 * lane - 256 * j] = acc[9 + 2 * j];
                    }
                }
                if (lane == 0) row[512] = acc[0];
                if (lane - 1u < 31u) row[1024 + lane] = 0.f;
            } else if (STOREM == 2) {
#pragma unroll
                for (int j = 0; j < 16; j++) row[lane + 64 * j] = acc[j];
                if (lane < 32u) row[1024 + lane] = lane ? 0.f : acc[0];
            } else if (STOREM == 3) {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    *(gf4 *)(row + 4 * (lane + 64 * j)) = v4f{acc[4 * j], acc[4 * j + 1], acc[4 * j + 2], acc[4 * j + 3]};
                if (lane < 32u) row[1024 + lane] = lane ? 0.f : acc[0];
            } else {
                float s = 0.f;
#pragma unroll
                for (int i = 0; i < 16; i++) s += acc[i];
                if (s == 12345.678f) row[lane] = s;
            }
        }
#if defined(NYQ_PLANE)
        if (STOREM == 1 && lane < f1 - f0) plane[f0 + lane] = nyq;
#endif
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// "Sweep" schedule (round 4): the same per-frame traffic and stand-in work, but the frames are dealt out so that everything
// in flight chip-wide is one compact window moving linearly through input and output (what the DRAM likes: r03 stream_shapes B)
// WITHOUT giving up the register reuse and without a workgroup barrier:
//   * a workgroup pulls GROUPS of SUBS x SUBF consecutive frames of one channel from the in-order device-wide queue (one
//     atomic per group: 7.5 k per launch instead of 90 k), one group ahead of its use;
//   * inside the workgroup the waves draw SUB-CHUNKS of SUBF = 4 frames from a ticket counter in LDS (dynamic: the waves of a
//     SIMD run at different speeds), so its 12 waves write 12 neighbouring 16 KB pieces at any time;
//   * the first frame of the NEXT sub-chunk (all 16 slots) is requested in the last frame of the current one, where the kept
//     slots are dead anyway: a sub-chunk start exposes no load latency and needs no extra registers.
// ---------------------------------------------------------------------------------------------------------------------
static int g_iters = 24;  // launches per measurement (mode 9: long loops for the power probe)
template <int FMA, int LDSR, int WAVES, int SUBS>
__global__ __launch_bounds__(64 * WAVES) __attribute__((amdgpu_waves_per_eu(3, 3))) void k_sweep(const float *wav_, float *spec_, uint32_t n_chan, uint32_t n_samples, uint32_t T,
                                                      uint32_t groups_per_chan, uint32_t *queue, float seed) {
    constexpr uint32_t SUBF = 4, GF = SUBS * SUBF, K = 8;
    // (the scheduler's words sit in front of the dynamic LDS: keep the slabs 16-byte aligned — misaligned ds_read_b128 cost
    // this skeleton 0.27 ms per launch before the attribute was there)
    extern __shared__ __attribute__((aligned(256))) v2f lds[];
    // (BEHIND the slabs: static __shared__ words are placed in front of the dynamic segment, which then starts at byte 72 whatever
    // alignment the extern array asks for — every ds_read_b128 of the stand-in exchange was misaligned by 8 bytes, 0.28 ms per
    // launch; the "ldsr 2" lines of profiles/r04_ubench_stft_skeleton_sweep.txt before this fix measure that, not the schedule)
    uint32_t *const sched = reinterpret_cast<uint32_t *>(lds + WAVES * 1100);
    uint32_t &s_ticket = sched[0];
    uint32_t *const s_group = sched + 1, *const s_tag = sched + 1 + K;
    const uint32_t lane = threadIdx.x & 63u, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    v2f *slab = lds + wave * 1100;
    const uint32_t n_groups = n_chan * groups_per_chan, fpc = T - 4;  // interior frames per channel
    if (threadIdx.x == 0) {
        s_ticket = 0;
        for (uint32_t i = 0; i < K; i++) s_tag[i] = 0xffffffffu;
        s_group[0] = blockIdx.x;  // sequence 0: static
        s_tag[0] = 0;
        s_group[1] = atomicAdd(queue, 1u) + gridDim.x;  // sequence 1
        s_tag[1] = 1;
    }
    __syncthreads();
    // resolves ticket t -> (valid, channel, first frame); fills the group slot one sequence ahead when t opens a group
    auto draw = [&](uint32_t &ch, uint32_t &f0, uint32_t &nf) -> int {  // 1 valid, 0 skip (ragged tail of a group), -1 no more groups
        uint32_t t = 0;
        // (explicit LDS address space everywhere: through a generic volatile pointer these become FLAT operations, which count on
        // vmcnt, and every draw then waits for the wave's outstanding row stores)
        if (lane == 0) t = __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t *)&s_ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        t = __builtin_amdgcn_readfirstlane(t);
        const uint32_t q = t / SUBS, sub = t % SUBS;
        if (sub == 0 && q >= 1) {  // this wave opens group q: pull the group of sequence q + 1
            uint32_t g2 = 0;
            if (lane == 0) g2 = atomicAdd(queue, 1u);
            g2 = __builtin_amdgcn_readfirstlane(g2) + gridDim.x;
            if (lane == 0) {
                // (LDS executes one wave's DS operations in order and is the only copy of these words: program order of the two
                // stores is all the release there is to do — a workgroup-scope fence would drain vmcnt, i.e. wait for the frame's
                // row stores)
                *(volatile __attribute__((address_space(3))) uint32_t *)&s_group[(q + 1) % K] = g2;
                asm volatile("" ::: "memory");
                *(volatile __attribute__((address_space(3))) uint32_t *)&s_tag[(q + 1) % K] = q + 1;
            }
        }
        for (uint32_t spin = 0; *(volatile __attribute__((address_space(3))) uint32_t *)&s_tag[q % K] != q; spin++) {
            if (spin > (1u << 22)) return -1;  // (cannot happen: the opener of group q - 1 fills the slot; bounded anyway)
            __builtin_amdgcn_s_sleep(2);
        }
        asm volatile("" ::: "memory");
        const uint32_t g = __builtin_amdgcn_readfirstlane(*(volatile __attribute__((address_space(3))) uint32_t *)&s_group[q % K]);
        if (g >= n_groups) return -1;
        ch = g / groups_per_chan;
        const uint32_t fr = (g % groups_per_chan) * GF + sub * SUBF;  // frame offset inside the channel's interior range
        if (fr >= fpc) return 0;
        f0 = 2 + fr;
        nf = min(SUBF, fpc - fr);
        return 1;
    };
    const uint32_t L = 4 * (lane & 15) + (lane >> 4);
    uint32_t ch = 0, f0 = 0, nf = 0;
    int st;
    do st = draw(ch, f0, nf); while (st == 0);
    if (st < 0) return;
    v2f x[16];
    float acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = seed + i;
    {
        const gf *wav = (const gf *)(wav_ + (size_t)ch * n_samples);
        const int64_t e0 = (int64_t)f0 * HOP - N_FFT / 2;
#pragma unroll
        for (int m = 0; m < 16; m++) x[m] = *(const gf2 *)(wav + e0 + 2 * (L + 64 * m));
    }
    // one frame: consume x, request what the next frame needs (KIND 0: the next frame of the sub-chunk, 4 new slots; 1: the first
    // frame of the next sub-chunk, all 16 slots; 2: nothing), stand-in work, row stores.  Straight-line code per KIND: with the
    // frames in a run-time loop the compiler cannot count the outstanding loads / stores across the back edge and waits for
    // vmcnt(0) at the loop head, i.e. for the previous frame's STORES (0.76 ms instead of 0.50 with the kernel's work).
    auto frame = [&](auto kind, const gf *wav, gf *spec, uint32_t f, const gf *wn, int64_t e0n) {
        constexpr int KIND = decltype(kind)::value;
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = acc[i] * 0.999f + (x[i].x + x[i].y);
        if constexpr (KIND == 0) {
            const int64_t e1 = (int64_t)(f + 1) * HOP - N_FFT / 2;
#pragma unroll
            for (int m = 0; m < 12; m++) x[m] = x[m + 4];
#pragma unroll
            for (int m = 12; m < 16; m++) x[m] = *(const gf2 *)(wav + e1 + 2 * (L + 64 * m));
        } else if constexpr (KIND == 1) {
#pragma unroll
            for (int m = 0; m < 16; m++) x[m] = *(const gf2 *)(wn + e0n + 2 * (L + 64 * m));
        }
#pragma unroll 1
        for (int r = 0; r < FMA; r++) {
#pragma unroll
            for (int i = 0; i < 16; i++) acc[i] = __builtin_fmaf(acc[i], 1.0001f, 0.25f);
        }
#pragma unroll 1
        for (int r = 0; r < LDSR; r++) {
            const uint32_t base = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)slab);
#define WA(i, o) "ds_write_addtid_b32 %" #i " offset:" #o "\n\t"
            asm volatile("s_mov_b32 m0, %16\n\ts_nop 0\n\t" WA(0, 0) WA(1, 272) WA(2, 544) WA(3, 816) WA(4, 1088) WA(5, 1360) WA(6, 1632) WA(7, 1904)
                         WA(8, 2176) WA(9, 2448) WA(10, 2720) WA(11, 2992) WA(12, 3264) WA(13, 3536) WA(14, 3808) WA(15, 4080)
                         :: "v"(acc[0]), "v"(acc[1]), "v"(acc[2]), "v"(acc[3]), "v"(acc[4]), "v"(acc[5]), "v"(acc[6]), "v"(acc[7]),
                         "v"(acc[8]), "v"(acc[9]), "v"(acc[10]), "v"(acc[11]), "v"(acc[12]), "v"(acc[13]), "v"(acc[14]), "v"(acc[15]),
                         "s"(base) : "memory");
            asm volatile("s_mov_b32 m0, %16\n\ts_nop 0\n\t" WA(0, 4352) WA(1, 4624) WA(2, 4896) WA(3, 5168) WA(4, 5440) WA(5, 5712) WA(6, 5984) WA(7, 6256)
                         WA(8, 6528) WA(9, 6800) WA(10, 7072) WA(11, 7344) WA(12, 7616) WA(13, 7888) WA(14, 8160) WA(15, 8432)
                         :: "v"(acc[15]), "v"(acc[14]), "v"(acc[13]), "v"(acc[12]), "v"(acc[11]), "v"(acc[10]), "v"(acc[9]), "v"(acc[8]),
                         "v"(acc[7]), "v"(acc[6]), "v"(acc[5]), "v"(acc[4]), "v"(acc[3]), "v"(acc[2]), "v"(acc[1]), "v"(acc[0]),
                         "s"(base) : "memory");
#undef WA
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // (explicit LDS address space: inside the lambda the slab pointer is "flat" to the compiler, and flat loads go through
            // the vector memory pipe)
            const __attribute__((address_space(3))) float *sf = (const __attribute__((address_space(3))) float *)slab;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const v4f v = *(const __attribute__((address_space(3))) v4f *)(&sf[(4 * i + (lane >> 4)) * 68 + 4 * (lane & 15)]);
                acc[2 * i] += v.x + v.y;
                acc[2 * i + 1] += v.z + v.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        gf *row = spec + (size_t)f * PITCH;
#pragma unroll
        for (int jj = 0; jj < 8; jj++) row[lane + 64 * jj] = acc[jj];
#pragma unroll
        for (int jj = 0; jj < 8; jj++) row[1024 - lane - 64 * jj] = acc[8 + jj];
        if (lane == 0) row[512] = acc[0];
        if (lane - 1u < 31u) row[1024 + lane] = 0.f;
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;
    for (;;) {
        const gf *wav = (const gf *)(wav_ + (size_t)ch * n_samples);
        gf *spec = (gf *)(spec_ + (size_t)ch * T * PITCH);
        uint32_t chn = 0, f0n = 0, nfn = 0;
        int stn;
        if (nf == SUBF) {
            frame(K0{}, wav, spec, f0, wav, 0);
            frame(K0{}, wav, spec, f0 + 1, wav, 0);
            // the next sub-chunk is drawn a frame before its first frame is requested: the ticket, a possible queue pull and
            // the table look-up are over when the last frame starts
            do stn = draw(chn, f0n, nfn); while (stn == 0);
            frame(K0{}, wav, spec, f0 + 2, wav, 0);
            const gf *wn = (const gf *)(wav_ + (size_t)chn * n_samples);
            const int64_t e0n = (int64_t)f0n * HOP - N_FFT / 2;
            if (stn > 0) frame(K1{}, wav, spec, f0 + 3, wn, e0n);
            else frame(K2{}, wav, spec, f0 + 3, wn, e0n);
        } else {  // ragged tail of a channel (1..3 frames)
            for (uint32_t j = 0; j + 1 < nf; j++) frame(K0{}, wav, spec, f0 + j, wav, 0);
            do stn = draw(chn, f0n, nfn); while (stn == 0);
            const gf *wn = (const gf *)(wav_ + (size_t)chn * n_samples);
            const int64_t e0n = (int64_t)f0n * HOP - N_FFT / 2;
            if (stn > 0) frame(K1{}, wav, spec, f0 + nf - 1, wn, e0n);
            else frame(K2{}, wav, spec, f0 + nf - 1, wn, e0n);
        }
        if (stn < 0) return;
        ch = chn;
        f0 = f0n;
        nf = nfn;
    }
}

template <int FMA, int LDSR, int WAVES, int SUBS>
static void run_sweep(const float *wav, float *spec, uint32_t n_chan, uint32_t n_samples, uint32_t T, uint32_t *q, int gap_us) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t gpc = (T - 4 + SUBS * 4 - 1) / (SUBS * 4);
    auto kern = k_sweep<FMA, LDSR, WAVES, SUBS>;
    const size_t lds = (size_t)WAVES * 1100 * sizeof(v2f) + 128;  // + the scheduler's words
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<float> ts;
    for (int i = 0; i < g_iters; i++) {
        hipMemsetAsync(q, 0, 4, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(256), dim3(64 * WAVES), lds, 0, wav, spec, n_chan, n_samples, T, gpc, q, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 4) ts.push_back(ms);
        if (gap_us) {
            hipDeviceSynchronize();
            struct timespec t = {0, gap_us * 1000};
            nanosleep(&t, nullptr);
        }
    }
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2], frames = (double)n_chan * (T - 4);
    printf("sweep: groups of %2d x 4 frames, fma %4d ldsr %d waves %2d: median %.3f ms min %.3f  %.0f GB/s\n", SUBS, FMA * 16, LDSR, WAVES, med, ts[0],
           frames * 6148.0 / med / 1e6);
    fflush(stdout);
}

static int g_json = 0;       // > 0: JSON output (mode 7), counts the entries printed
static uint32_t g_grid = 0;  // 0: one workgroup per CU (persistent); else this many workgroups (argv[3])
template <int LOADM, int STOREM, int FMA, int LDSR, int WAVES, int LDSM = 0>
static void run(const char *name, const float *wav, float *spec, uint32_t n_chan, uint32_t n_samples, uint32_t T, uint32_t *q,
                int gap_us) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const uint32_t cpc = (T - 4 + CHUNK - 1) / CHUNK;
    auto kern = k<LOADM, STOREM, FMA, LDSR, WAVES, LDSM>;
    const size_t lds = (size_t)WAVES * 1100 * sizeof(v2f) + 128;  // + the scheduler's words
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    std::vector<float> ts;
    for (int i = 0; i < g_iters; i++) {
        hipMemsetAsync(q, 0, 4, 0);
        hipEventRecord(e0);
        hipLaunchKernelGGL(kern, dim3(g_grid ? g_grid : 256), dim3(64 * WAVES), lds, 0, wav, spec, n_chan, n_samples, T, cpc, q, 1.0f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (i >= 4) ts.push_back(ms);
        if (gap_us) {
            hipDeviceSynchronize();
            struct timespec t = {0, gap_us * 1000};
            nanosleep(&t, nullptr);
        }
    }
    std::sort(ts.begin(), ts.end());
    const double med = ts[ts.size() / 2];
    const double frames = (double)n_chan * (T - 4);
    const double bytes = frames * ((LOADM ? 2048.0 : 0.0) + (STOREM ? 4100.0 : 0.0));
    if (g_json) {
        printf("%s{\"load\": %d, \"store\": %d, \"fma_per_frame\": %d, \"lds_rounds\": %d, \"waves\": %d, \"median_ms\": %.4f, \"min_ms\": %.4f, \"p90_ms\": %.4f}",
               g_json++ > 1 ? ", " : "", LOADM, STOREM, FMA * 16, LDSR, WAVES, med, ts[0], ts[ts.size() * 9 / 10]);
        return;
    }
    printf("%-4s ldsm %d load %d store %d fma %4d ldsr %d waves %2d: median %.3f ms min %.3f  %.0f GB/s\n", name, LDSM, LOADM, STOREM, FMA * 16,
           LDSR, WAVES, med, ts[0], bytes / med / 1e6);
    fflush(stdout);
}

int main(int argc, char **argv) {
    const int gap_us = argc > 1 ? atoi(argv[1]) : 0;
    if (argc > 3 && (argc < 3 || atoi(argv[2]) != 9)) g_grid = (uint32_t)atoi(argv[3]);
    const uint32_t n_chan = 128, n_samples = 1440000, T = n_samples / HOP + 1;
    float *wav, *spec;
    uint32_t *q;
    hipMalloc(&wav, (size_t)n_chan * n_samples * 4);
    hipMalloc(&spec, (size_t)n_chan * T * (PITCH + 1) * 4);  // (+ 1: the Nyquist plane of the NYQ_PLANE build)
    hipMalloc(&q, 4);
    {   // pseudo-random input (bit toggling matters on a power-limited part)
        std::vector<float> h((size_t)n_samples);
        uint32_t s = 12345;
        for (auto &v : h) {
            s = s * 1664525u + 1013904223u;
            v = ((int32_t)s) * (0.3f / 2147483648.0f);
        }
        for (uint32_t c = 0; c < n_chan; c++) hipMemcpy(wav + (size_t)c * n_samples, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    }
#define R(L, S, F, D, W, M) run<L, S, F, D, W, M>("", wav, spec, n_chan, n_samples, T, q, gap_us)
    if (argc > 2 && atoi(argv[2]) == 7) {
        // bench.py's `roofline.memory_skeleton`: one JSON line — the kernel's own access structure (load shape 3, store
        // shape 1, 12 waves per CU) with no arithmetic, and with the kernel's amount of stand-in work
        g_json = 1;
        printf("{\"gap_us\": %d, \"workload\": \"128 channels x 1440000 samples, n_fft 2048 / hop 512, rows at pitch 1056\", \"runs\": [", gap_us);
        R(3, 1, 0, 0, 12, 2);
        R(3, 1, 42, 2, 12, 2);
        printf("]}\n");
        return 0;
    }
    printf("# gap between launches: %d us\n", gap_us);
    if (argc > 2 && atoi(argv[2]) == 10) {  // row layout A/B (build with and without -DNYQ_PLANE): stores alone, skeleton, with the kernel's work
        printf("# rows at pitch %u floats%s\n", PITCH, PITCH == 1024 ? " + Nyquist plane" : "");
        for (int rep = 0; rep < 3; rep++) {
            R(0, 1, 0, 0, 12, 2);
            R(3, 1, 0, 0, 12, 2);
            R(3, 1, 42, 2, 12, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 9) {
        // power probe (scripts/power_probe.sh): argv[3] = 0: the kernel's own schedule, 1: the sweep schedule, both with the kernel's
        // amount of stand-in work, back to back for about ten seconds
        g_iters = 20000;
        const int which = argc > 3 ? atoi(argv[3]) : 0;
        g_grid = 0;
        if (which == 0) R(3, 1, 42, 2, 12, 2);
        else run_sweep<42, 2, 12, 6>(wav, spec, n_chan, n_samples, T, q, gap_us);
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 1) {
        // store shapes next to the kernel's amount of VALU / LDS work: the kernel's pattern (1), 16 aligned dword stores
        // (2), four 16-byte stores (3; with a third LDS round standing for the transposition that would feed them),
        // adjacent-bin pairs (4)
        for (int rep = 0; rep < 3; rep++) {
            R(3, 1, 42, 2, 12, 2);
            R(3, 2, 42, 2, 12, 2);
            R(3, 3, 42, 2, 12, 2);
            R(3, 3, 42, 3, 12, 2);
            R(3, 4, 42, 2, 12, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 8) {  // the sweep schedule against the kernel's schedule: no work / the kernel's amount of work
        for (int rep = 0; rep < 2; rep++) {
            R(3, 1, 0, 0, 12, 2);
            run_sweep<0, 0, 12, 12>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<0, 0, 12, 24>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<0, 0, 12, 6>(wav, spec, n_chan, n_samples, T, q, gap_us);
            R(3, 1, 42, 2, 12, 2);
            run_sweep<42, 2, 12, 12>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<42, 2, 12, 24>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<42, 2, 12, 6>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<30, 2, 12, 12>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<42, 0, 12, 12>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<0, 2, 12, 12>(wav, spec, n_chan, n_samples, T, q, gap_us);
            run_sweep<10, 0, 12, 12>(wav, spec, n_chan, n_samples, T, q, gap_us);
            R(3, 1, 42, 0, 12, 2);
            R(3, 1, 0, 2, 12, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 6) {  // grid-size runs (argv[3]): stores alone in two shapes, skeleton, with work; 4-wave workgroups too
        for (int rep = 0; rep < 2; rep++) {
            R(0, 1, 0, 0, 12, 2);
            R(0, 3, 0, 0, 12, 2);
            R(3, 1, 0, 0, 12, 2);
            R(3, 1, 42, 2, 12, 2);
            R(0, 3, 0, 0, 4, 2);
            R(3, 1, 0, 0, 4, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 5) {  // with the kernel's amount of work: waves per CU, 16-byte loads / stores
        for (int rep = 0; rep < 3; rep++) {
            R(3, 1, 42, 2, 12, 2);
            R(3, 1, 42, 2, 8, 2);
            R(3, 1, 42, 2, 10, 2);
            R(2, 1, 42, 2, 12, 2);
            R(2, 3, 42, 2, 12, 2);
            R(2, 3, 42, 2, 8, 2);
            R(2, 3, 42, 2, 10, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 4) {  // chunk-length builds (-DCHUNK_FRAMES=n): stores alone, memory skeleton, with the kernel's work
        printf("# chunk %u frames\n", CHUNK);
        for (int rep = 0; rep < 2; rep++) {
            R(0, 1, 0, 0, 12, 2);
            R(3, 1, 0, 0, 12, 2);
            R(3, 1, 42, 2, 12, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 3) {
        // the memory skeleton alone (no VALU / LDS work): load shapes x store shapes x waves per CU
        for (int rep = 0; rep < 2; rep++) {
            R(3, 1, 0, 0, 12, 2);
            R(1, 1, 0, 0, 12, 2);
            R(2, 1, 0, 0, 12, 2);
            R(3, 2, 0, 0, 12, 2);
            R(3, 3, 0, 0, 12, 2);
            R(2, 3, 0, 0, 12, 2);
            R(3, 1, 0, 0, 8, 2);
            R(3, 1, 0, 0, 16, 2);
            R(2, 3, 0, 0, 16, 2);
            R(0, 1, 0, 0, 12, 2);
            R(0, 3, 0, 0, 12, 2);
            R(3, 0, 0, 0, 12, 2);
        }
        return 0;
    }
    if (argc > 2 && atoi(argv[2]) == 2) {
        // what an amount of VALU or LDS work is worth in launch time, next to the kernel's memory traffic (power-capped part)
        for (int rep = 0; rep < 2; rep++) {
            R(3, 1, 42, 2, 12, 2);
            R(3, 1, 36, 2, 12, 2);
            R(3, 1, 30, 2, 12, 2);
            R(3, 1, 21, 2, 12, 2);
            R(3, 1, 42, 1, 12, 2);
            R(3, 1, 42, 0, 12, 2);
            R(3, 1, 0, 2, 12, 2);
            R(3, 1, 0, 0, 12, 2);
            R(3, 1, -42, 2, 12, 2);
        }
        return 0;
    }
    R(0, 0, 48, 0, 12, 0);
    R(0, 0, -48, 0, 12, 0);
    R(0, 0, 48, 2, 12, 2);
    R(0, 0, -48, 2, 12, 2);
    R(3, 1, 48, 2, 12, 2);
    R(3, 1, -48, 2, 12, 2);
    R(3, 1, 44, 2, 12, 2);
    R(3, 1, -44, 2, 12, 2);
    R(3, 1, 48, 2, 12, 2);
    R(3, 1, -48, 2, 12, 2);
    return 0;
}
