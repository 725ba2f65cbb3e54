// kernels_stft_long.hip — n_fft 32768 as SIXTEEN WAVE TRANSFORMS plus one combining pass (round 5).
//
// stft_block_kernel (kernels_stft.hip / stft_block.h) runs the long transforms as four radix-16 Stockham passes of one workgroup
// with the whole frame in LDS between them: ten workgroup barriers per frame, 512 threads (two waves per SIMD), nothing to
// overlap a barrier or a fetch with — 18 us per frame where its arithmetic is 4 (profiles/r04_block_phase_prof.txt: barriers
// 34 %, fetch 23 %; 0.17 of the HBM roofline).  Here the same 16384-point complex transform is decimated in time ONCE,
//
//     Z[k2 + 1024 k1] = sum_{n1 < 16} W_16384^(n1 k2) W_16^(n1 k1)  S_n1[k2],     S_n1 = DFT_1024 of z[n1 + 16 n2],
//
// and the sixteen 1024-point transforms S_n1 are exactly what the n_fft 2048 wave kernel does per frame: wave n1 of the
// workgroup runs WaveFft<10>'s three register passes with its two plane exchanges through a slab of its own — no workgroup
// barrier, sixteen waves (four per SIMD) drifting past each other as in the wave kernel.  What needs the workgroup:
//   (0) the frame reaches the slabs transposed: thread t loads the windowed points t + 1024 j (coalesced), all of which belong
//       to sub-transform t mod 16, column t / 16 — precisely the sixteen inputs one lane of that wave's first pass owns;
//   (2) the combining pass: thread t = k2 reads S_n1[t] from the sixteen slabs and runs BlockFft<14>'s last pass (the same
//       twiddled radix-16 FMA butterfly, ten constants per thread);
//   (3) the real-FFT split pass needs Z[Nc - k]: Z goes through LDS once more (over the slabs), as in the block kernel.
// Five barriers per frame, each between phases that every wave reaches at about the same time.  Arithmetic, twiddles, window and
// output are the block kernel's (same tables, same butterflies): the results agree with it to rounding.
#include <hip/hip_runtime.h>

#include <type_traits>
#include <vector>

#include "kernels.h"
#include "stft_block.h"
#include "stft_core.h"
#include "stft_wave.h"

namespace th {

namespace {

__device__ __forceinline__ float nmin_l(float a, float b) { return fminf(a, b); }
__device__ __forceinline__ float nmax_l(float a, float b) { return fmaxf(a, b); }
__device__ __forceinline__ float wave_min_l(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nmin_l(v, __shfl_xor(v, o, 64));
    return v;
}
__device__ __forceinline__ float wave_max_l(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = nmax_l(v, __shfl_xor(v, o, 64));
    return v;
}
// orders this wave's LDS writes before its later LDS reads for the compiler (the hardware runs one wave's DS operations in order)
__device__ __forceinline__ void wave_lds_sync_l() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup barrier for LDS traffic only.  __syncthreads() is a release / acquire fence over ALL address spaces: s_waitcnt vmcnt(0)
// in front of the s_barrier, i.e. every wave waits for its row stores of the previous frame (and for the next frame's samples) to
// complete before it may even arrive — thousands of cycles per barrier (scripts/subwave_prof.py).  The phases below only hand LDS
// data to each other; global memory is never shared between the threads of a launch.
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

constexpr int SUB_LOG2 = 10;                       // the sub-transform: WaveFft<10>, 1024 complex points, one wave
using SW = WaveFft<SUB_LOG2>;
constexpr int SUB_STRIDE = SW::SLAB_LEN + 2;       // cf32 per slab: 16-byte aligned, and consecutive slabs four banks apart (phase 0's scatter)

}  // namespace

// LOG2_R = 4 (n_fft 32768: sixteen sub-transforms, 1024 threads, one workgroup per CU), 3 (n_fft 16384: eight, 512 threads, two
// workgroups per CU) or 2 (n_fft 8192: four, 256 threads, four per CU).  The combining pass is the radix-R twiddled FMA butterfly of
// stft_wave.h / stft_block.h; with R < 16 a thread combines 16 / R bins k2 (tid, tid + 64 R, ...), so that it always ends up with the
// sixteen values Z[tid + 64 R c] the block plan's split pass expects.
template <int LOG2_R>
struct SubwaveCfg {
    static constexpr int R = 1 << LOG2_R, NT = 64 * R, KPT = 16 / R;   // sub-transforms = waves, threads, bins k2 per thread
    static constexpr int NTWC = R == 16 ? 10 : KPT * (R == 8 ? 4 : 2);  // combining-pass constants per thread
    static constexpr int WG_PER_CU = 16 / R;
    static constexpr size_t LDS = sizeof(cf32) * (SW::T2_LEN + SW::T3_LEN + (size_t)R * SUB_STRIDE) + 2 * R * sizeof(float);
};
template <int LOG2_R, bool AMP, bool REUSE>
// (HIP's second __launch_bounds__ argument is the least number of waves per SIMD to leave room for: four — 16 waves per CU, 128 VGPRs —
// at every R; without it the R = 8 kernel with resident samples took 149 registers and its second workgroup no longer fitted the CU)
__global__ __launch_bounds__(SubwaveCfg<LOG2_R>::NT, 4) void stft_subwave_kernel(
    StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab, uint32_t n_tiles, const cf32 *__restrict__ wtab_g,
    const cf32 *__restrict__ tw, const cf32 *__restrict__ twc, float *__restrict__ minmax) {
    using C = SubwaveCfg<LOG2_R>;
    using B = BlockFft<10 + LOG2_R>;
    constexpr int NC = B::NC, T = B::T, R = C::R, NT = C::NT, KPT = C::KPT;
    static_assert(T == NT && NC == R * SW::NC, "R 1024-point sub-transforms; the block plan's thread count");
    static_assert(SW::PLANES && SW::PAIRED && SW::NQ == 2 && SW::R3 == 4, "the n_fft 2048 plane plan");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf32 *const t2 = reinterpret_cast<cf32 *>(smem_raw);
    cf32 *const t3 = t2 + SW::T2_LEN;
    float *const red = reinterpret_cast<float *>(t3 + SW::T3_LEN);  // 2 x R floats
    cf32 *const slabs = reinterpret_cast<cf32 *>(red + 2 * R);       // R slabs; Z (Nc + 1 slots) lies over them in phase 3
    static_assert((SW::T2_LEN + SW::T3_LEN) % 2 == 0 && SUB_STRIDE % 2 == 0 && (2 * R) % 4 == 0, "slabs 16-byte aligned");
    static_assert(R * SUB_STRIDE >= NC + 1, "Z fits over the slabs");

    const uint32_t t = threadIdx.x, lane_w = t & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(t >> 6);
    // Persistent workgroups: chunk blockIdx.x, then every gridDim.x-th (the chunks of a launch are equally long but for the
    // channels' tails).  A workgroup of 1024 threads with 145 KB of LDS starts only when its predecessor on the CU has gone, and its
    // set-up (tables, first frame) is three dependent memory round trips: 34 of the 142 us an 8-frame chunk took (scripts/subwave_prof.py).
    uint32_t ct = blockIdx.x;
    if (ct >= n_tiles) return;

    // sub-transform tables: W_2048^i = tw[R i] (tw = W_{n_fft}^i, n_fft = 2 Nc = 2048 R)
    constexpr uint32_t TS = (uint32_t)(NC / SW::NC);
    for (uint32_t i = t; i < (uint32_t)SW::T2_LEN; i += (uint32_t)NT) t2[i] = tw[TS * SW::t2_index(i / SW::NS2, i % SW::NS2)];
    for (uint32_t i = t; i < (uint32_t)SW::T3_LEN; i += (uint32_t)NT) {
        const uint32_t r = i / SW::NS3 + 1, k = i % SW::NS3;
        t3[i] = tw[TS * ((r * k) * (2u * SW::NC / (SW::NS3 * SW::R3)))];
    }
    const cf32 stw_t = tw[t];  // the split twiddle of bin t
    __syncthreads();

    cf32 *const slab = slabs + (size_t)wv * SUB_STRIDE;
    // the frame stream of this workgroup: (chunk, frame) -> next frame of the chunk, else the first frame of the next chunk
    struct Cur {
        uint32_t ct, f, f1, spec_pitch;
        gptr<const float> wav;
        gptr<float> spec;
        bool valid;
    };
    auto open_chunk = [&](uint32_t c) -> Cur {
        Cur k{};
        k.valid = c < n_tiles;
        if (k.valid) {
            const uint32_t chan = chunk_tab[2 * (size_t)c];
            k.ct = c;
            k.f = chunk_tab[2 * (size_t)c + 1];
            k.f1 = min(k.f + g.frames_per_tile, jobs[chan].f_end);
            k.spec_pitch = jobs[chan].spec_pitch;
            k.wav = as_global(jobs[chan].wav);
            k.spec = as_global(jobs[chan].spec);
        }
        return k;
    };
    auto next_of = [&](const Cur &c) -> Cur {
        if (c.f + 1 < c.f1) {
            Cur k = c;
            k.f = c.f + 1;
            return k;
        }
        return open_chunk(c.ct + gridDim.x);
    };
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    // (addresses as "uniform base of piece j (SGPRs) + 8 tid": no per-thread 64-bit pointers to keep or spill)
    // raw samples (points tid + 1024 j) and window pairs of the frame being staged: locals of ONE loop iteration (loop-carried they
    // would stay live through the sub-transform: the first build spilled them to scratch right behind their loads)
    auto fetch = [&](const Cur &c, uint32_t tid, cf32 (&x)[16], auto j0_tag) {
        constexpr int J0 = decltype(j0_tag)::value;  // first piece to load (REUSE: 12 — the pieces below it moved down from the previous frame)
        const int64_t e0 = (int64_t)c.f * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;  // interior frames: the whole span is inside the channel
#pragma unroll
        for (int j = J0; j < 16; j++) {
#if !(defined(TH_SUBW_ABL) && (TH_SUBW_ABL & 2))  // ablation build: no sample loads
            const gptr<const float> pj = c.wav + (e0 + 2 * NT * (int64_t)j);
            x[j] = {pj[2u * tid], pj[2u * tid + 1u]};
#else
            x[j] = {0.25f, -0.125f};
#endif
        }
    };
    auto fetch_window = [&](uint32_t tid, cf32 (&xw)[16]) {
#pragma unroll
        for (int j = 0; j < 16; j++) {
#if defined(TH_SUBW_ABL) && (TH_SUBW_ABL & 1)  // ablation build: no window loads
            xw[j] = {0.5f, 0.25f};
#else
            xw[j] = (wtab_g + NT * j)[tid];
#endif
        }
    };
    // phase 0: window, then to sub-transform tid mod R as column tid / R (slot 64 j + column)
    auto stage = [&](uint32_t tid, const cf32 (&x)[16], const cf32 (&xw)[16]) {
        // column c of sub-transform n1 goes where the lane that owns it (lane_col(l) = c: l = (c >> 2) + 16 (c & 3)) reads in lane
        // order — reading by column was a 4-way bank conflict on all sixteen reads (SQ_LDS_BANK_CONFLICT 16 % of the LDS cycles);
        // R = 16: rotated by n1 within the 64 slots, so that the sixteen slabs a wave scatters to start on different bank pairs
        const uint32_t c = tid >> LOG2_R, n1 = tid & (uint32_t)(R - 1);
        cf32 *const dst = slabs + (size_t)n1 * SUB_STRIDE + (((c >> 2) + 16u * (c & 3u) - (R == 16 ? n1 : 0u)) & 63u);
#pragma unroll
        for (int j = 0; j < 16; j++) lds_st(&dst[64 * j], cf32{x[j].re * xw[j].re, x[j].im * xw[j].im});
    };
    Cur cur = open_chunk(ct);
    // REUSE (hop = n_fft / 4): the raw samples stay in registers from frame to frame — frame f + 1 is frame f moved down by four of
    // the thread's sixteen pieces, so only the new hop is requested (a quarter of the sample traffic; the re-read three quarters
    // came from the L2 / MALL at best: removing the sample loads altogether is worth 0.32 of 1.48 ms, profiles/r05_ab_subwave.txt)
    cf32 xk[REUSE ? 16 : 1];
    {
        cf32 x0[16], w0[16];
        fetch(cur, t & (uint32_t)(NT - 1), x0, std::integral_constant<int, 0>{});
        fetch_window(t & (uint32_t)(NT - 1), w0);
        stage(t & (uint32_t)(NT - 1), x0, w0);
        if constexpr (REUSE) {
#pragma unroll
            for (int j = 0; j < 16; j++) xk[j % (REUSE ? 16 : 1)] = x0[j];
        }
    }
    // The instruction arbiter serves the oldest wave of a SIMD first: of the four waves of a SIMD the first finishes a phase at half
    // time and the last one runs alone at the end, at a lone wave's issue rate.  Between two barriers a wave therefore LOWERS its
    // priority as it advances: whoever is behind goes first, and the four arrive together.  Measured: 1.476 ms with, 1.464 without
    // (profiles/r05_ab_subwave.txt) — off; -DTH_SUBW_PRIO=1 builds it.
#if !defined(TH_SUBW_REUSE)
#define TH_SUBW_REUSE 1
#endif
#if !defined(TH_SUBW_NT)
#define TH_SUBW_NT 1
#endif
#if TH_SUBW_NT
#define TH_SUBW_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define TH_SUBW_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
#if !defined(TH_SUBW_PRIO)
#define TH_SUBW_PRIO 0
#endif
#if TH_SUBW_PRIO
#define TH_PRIO(P) __builtin_amdgcn_s_setprio(P)
#else
#define TH_PRIO(P) do { } while (0)
#endif
#if defined(TH_SUBW_PROF)  // development instrumentation: shader-clock ticks per phase, summed over the workgroup's frames (thread 0's view)
    uint64_t prof[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp = __builtin_amdgcn_s_memtime();
#define TH_STAMP(I) do { const uint64_t now_ = __builtin_amdgcn_s_memtime(); prof[I] += now_ - stamp; stamp = now_; } while (0)
#else
#define TH_STAMP(I) do { } while (0)
#endif
    while (cur.valid) {
        // per-frame opaque copies of the thread / lane ids: every LDS address below is "f(id) + immediate"; left loop-invariant
        // they are hoisted out of the frame loop and spilled (see wave_frame)
        uint32_t tt = t, lane = lane_w;
        asm volatile("" : "+v"(tt), "+v"(lane));
        tt &= (uint32_t)(NT - 1);
        lane &= 63u;
        TH_STAMP(0);
        lds_barrier();  // (1) every slab holds its sub-transform's input (staged behind the previous frame's barrier 5)
        TH_STAMP(1);
        // the combining pass's ten constants (twiddle index k2 = tt): requested now, they land during the sub-transform
        // (from the plan's per-thread table [constant][thread]: ten coalesced loads — read straight from tw they are gathers with
        // strides of 16 .. 128 bytes, up to 64 cache lines per wave-load, and were the kernel's largest vector-memory client)
        cf32 wC[C::NTWC];
#pragma unroll
        for (int e = 0; e < C::NTWC; e++) wC[e] = (twc + NT * e)[tt];
        // ---- phase 1: the 1024-point transform of this wave (stft_wave.h; no workgroup barrier inside)
        cf32 z[16];
        {
            static_assert(SW::PLANES, "lane_col(l) = 4 (l & 15) + (l >> 4): the staging layout above is its inverse");
            const uint32_t at = (lane - (R == 16 ? wv : 0u)) & 63u;
#pragma unroll
            for (int m = 0; m < 16; m++) z[m] = lds_ld(&slab[64 * m + at]);
        }
        wave_lds_sync_l();
        cf32 za[SW::NQ][SW::R3], zb[SW::NQ][SW::R3];
        {
            cf32 w2[SW::NT2];
            SW::load_t2(lane, w2, t2);
#if defined(TH_SUBW_ABL) && (TH_SUBW_ABL & 4)  // ablation build: no sub-transform
            if (g.hop == 0x7fffffffu)
#endif
            {
            TH_PRIO(3);
            SW::pass1(lane, z, slab);
            wave_lds_sync_l();
            TH_PRIO(2);
            SW::read1(lane, z, slab);
            wave_lds_sync_l();
            SW::pass2_w(lane, z, w2, slab);
            wave_lds_sync_l();
            TH_PRIO(1);
            cf32 wa[SW::NQ][SW::NT3], wb[SW::NQ][SW::NT3];
            const typename SW::PairBase pbs = SW::pair_base(lane);
            SW::load_t3_paired(pbs, wa, wb, t3);
            SW::read2_paired(lane, pbs, za, zb, slab);
            wave_lds_sync_l();
            SW::pass3_paired_w(za, zb, wa, wb);
            }
        }
        // S_wv[k2] to the wave's own slab in natural order: za[q][r] = S[A_q + 256 r], zb[q][r] = S[B_q + 256 r]
#pragma unroll
        for (int q = 0; q < SW::NQ; q++) {
            const uint32_t a = SW::jj_a(lane, q), b = SW::jj_b(lane, q);
#pragma unroll
            for (int r = 0; r < SW::R3; r++) {
                lds_st(&slab[a + (uint32_t)(r * SW::NS3)], za[q][r]);
                lds_st(&slab[b + (uint32_t)(r * SW::NS3)], zb[q][r]);
            }
        }
        TH_PRIO(0);
        TH_STAMP(2);
        lds_barrier();  // (2) all sixteen spectra are in the slabs
        TH_STAMP(3);
        // the next frame of the stream (the next chunk's first when this chunk ends): its samples travel during the combining pass
        // and the Z exchange
        // (unconditional register flow: behind the stream's last frame the same frame is staged once more — in bounds, never read)
        const Cur nxt = next_of(cur);
        cf32 x[16], xw[16];
        if constexpr (REUSE) {
            // the next frame continues this chunk: four pieces down, the new hop requested; a new chunk: everything
            if (nxt.valid && nxt.ct == cur.ct) {  // workgroup-uniform
#pragma unroll
                for (int j = 0; j < 12; j++) x[j] = xk[(j + 4) % (REUSE ? 16 : 1)];
                fetch(nxt, tt, x, std::integral_constant<int, 12>{});
            } else {
                fetch(nxt.valid ? nxt : cur, tt, x, std::integral_constant<int, 0>{});
            }
        } else {
            fetch(nxt.valid ? nxt : cur, tt, x, std::integral_constant<int, 0>{});
        }
        // ---- phase 2: the combining pass, thread = k2
        if constexpr (R == 16) {
#pragma unroll
            for (int n1 = 0; n1 < 16; n1++) z[n1] = lds_ld(&slabs[(size_t)n1 * SUB_STRIDE + tt]);
#if defined(TH_SUBW_ABL) && (TH_SUBW_ABL & 8)  // ablation build: no combining butterfly
            if (g.hop == 0x7fffffffu)
#endif
            B::pass_last(z, wC);  // z[c] = Z[tt + 1024 c]
        } else {
            // bin k2 = tt + NT u: Z[k2 + 1024 k1] = radix-R butterfly over n1 of S_n1[k2] with t = W_Nc^k2 = slot c = u + KPT k1
            cf32 v[KPT][R];
#pragma unroll
            for (int u = 0; u < KPT; u++)
#pragma unroll
                for (int n1 = 0; n1 < R; n1++) v[u][n1] = lds_ld(&slabs[(size_t)n1 * SUB_STRIDE + tt + (uint32_t)(NT * u)]);
#pragma unroll
            for (int u = 0; u < KPT; u++) {
                if constexpr (R == 8) {
                    cf32 b[8];
#pragma unroll
                    for (int n1 = 0; n1 < 8; n1++) b[n1] = v[u][n1 % R];
                    SW::bfly8_tw(b, wC[(4 * u) % C::NTWC], wC[(4 * u + 1) % C::NTWC], wC[(4 * u + 2) % C::NTWC], wC[(4 * u + 3) % C::NTWC]);  // natural order in and out
#pragma unroll
                    for (int k1 = 0; k1 < 8; k1++) z[(u + KPT * k1) % 16] = b[k1];
                } else {
                    cf32 a = v[u][0], b = v[u][1 % R], c = v[u][2 % R], d = v[u][3 % R];
                    bfly4_tw(a, b, c, d, wC[(2 * u) % C::NTWC], wC[(2 * u + 1) % C::NTWC]);  // out: a = X0, c = X1, b = X2, d = X3
                    z[(u + KPT * 0) % 16] = a;
                    z[(u + KPT * 1) % 16] = c;
                    z[(u + KPT * 2) % 16] = b;
                    z[(u + KPT * 3) % 16] = d;
                }
            }
        }
        TH_STAMP(4);
        lds_barrier();  // (3) the slabs have been read: Z may go over them
        fetch_window(tt, xw);  // (behind the combining pass: its ten constants are dead, 32 registers for the window pairs)
        B::write_z(tt, z, slabs);
        lds_barrier();  // (4)
        cf32 zm[8];
        B::split_read(tt, slabs, zm);
        lds_barrier();  // (5) the slabs are free again
        TH_STAMP(5);
        // phase 0 of the NEXT frame goes in front of this frame's split pass: its latency (samples, window) is behind us, and the
        // row stores below overlap the next sub-transform (lds_barrier does not wait for them)
        stage(tt, x, xw);
        if constexpr (REUSE) {
#pragma unroll
            for (int j = 0; j < 16; j++) xk[j % (REUSE ? 16 : 1)] = x[j];
        }
        const gptr<float> row = cur.spec + (size_t)cur.f * cur.spec_pitch;
        // (non-temporal row stores, TH_SUBW_NT: the 64 KB a frame writes should not push the 96 KB of samples the next frame re-reads out of the L2)
        auto emit = [&](uint32_t k, float p) {
            if constexpr (AMP) {
                TH_SUBW_STORE(&row[k], power_to_amp(p));
            } else {
                const float d = power_to_dB(p);
                TH_SUBW_STORE(&row[k], d);
                lmin = nmin_l(lmin, d);
                lmax = nmax_l(lmax, d);
            }
        };
#if defined(TH_SUBW_ABL) && (TH_SUBW_ABL & 16)  // ablation build: no split pass, no rows
        if (g.hop == 0x7fffffffu)
#endif
        B::split_compute(tt, z, zm, stw_t, emit);
        {   // complete the row's last 128-byte line (see wave_frame)
            const uint32_t height = (uint32_t)(NC + 1), padn = cur.spec_pitch - height;
            if (tt - 1u < ((padn < 32u && cur.spec_pitch % 32u == 0) ? padn : 0u)) row[height - 1u + tt] = 0.0f;
        }
        TH_STAMP(6);
        if (!(nxt.valid && nxt.ct == cur.ct)) {  // the chunk ends: its (min, max) pair, folded per channel by wave_post_kernel
#if defined(TH_SUBW_PROF)
            if (minmax != nullptr && t == 0) {  // (instead: phase TH_SUBW_PROF's ticks (7: all phases) per chunk so far)
                uint64_t tot = 0;
                for (int i = 0; i < 8; i++) tot += prof[i];
                minmax[2 * (size_t)cur.ct] = (float)tot;
                minmax[2 * (size_t)cur.ct + 1] = (float)(TH_SUBW_PROF < 7 ? prof[TH_SUBW_PROF] : tot);
                for (int i = 0; i < 8; i++) prof[i] = 0;
            }
#else
            if (minmax != nullptr) {
                const float a = wave_min_l(lmin), b = wave_max_l(lmax);
                if (lane_w == 0) {
                    red[2 * wv] = a;
                    red[2 * wv + 1] = b;
                }
                lds_barrier();  // (red[] is rewritten at the next chunk's end, at least five barriers from here)
                if (t == 0) {
                    float mn = red[0], mx = red[1];
                    for (int w = 1; w < R; w++) {
                        mn = nmin_l(mn, red[2 * w]);
                        mx = nmax_l(mx, red[2 * w + 1]);
                    }
                    minmax[2 * (size_t)cur.ct] = mn;
                    minmax[2 * (size_t)cur.ct + 1] = mx;
                }
                lmin = __builtin_inff();
                lmax = -__builtin_inff();
            }
#endif
        }
        cur = nxt;
    }
}

// ------------------------------------------------------------------------------------------
// n_fft 65536 (Nc = 32768 packed points) on the same machinery: one radix-2 decimation-in-FREQUENCY step in front,
//     a[n] = z[n] + z[n + 16384],   b[n] = (z[n] - z[n + 16384]) W_32768^n,      Z[2 k] = DFT_16384(a)[k],  Z[2 k + 1] = DFT_16384(b)[k],
// and the sixteen-wave pipeline above (staging -> sixteen 1024-point wave transforms -> radix-16 combining pass) run twice per frame,
// once on a and once on b.  Both half-spectra end up as thread t's A[t + 1024 c] and B[t + 1024 c] — and the real-FFT split pass
// never mixes them: bin m = 2 k pairs Z[2 k] = A[k] with Z[N - 2 k] = A[16384 - k], bin m = 2 k + 1 pairs B[k] with B[16383 - k].
// So A goes through the n_fft 32768 plan's mirror exchange and split pass unchanged (its bins k are the even bins 2 k: W_65536^(2 k) =
// W_32768^k), B through a plain reversal with the twiddles W_65536^(2 k + 1), and a thread's results are the adjacent bins (2 k, 2 k + 1)
// and (N - 2 k - 1, N - 2 k): 8-byte stores.  Eleven barriers per frame where the planar block plan has 33; 1024 threads instead of 512.
// ------------------------------------------------------------------------------------------
// (non-temporal row stores: 2.50 -> 2.44 ms.  Measured and not adopted: the first pass leaving lo - hi of every point in a per-workgroup
// 128 KB stash in global memory for the second pass to read back instead of loading samples and window pairs twice — 2.96 ms: the stash
// stores and the L1 invalidate in front of the read-back cost more than the 384 KB of L2 reads they save)
#if !defined(TH_SUBW2_NT)
#define TH_SUBW2_NT 1
#endif
#if TH_SUBW2_NT
#define TH_SUBW2_STORE(PTR, VAL) __builtin_nontemporal_store((VAL), (PTR))
#else
#define TH_SUBW2_STORE(PTR, VAL) (*(PTR) = (VAL))
#endif
template <bool AMP>
__global__ __launch_bounds__(1024, 4) void stft_subwave2_kernel(StftGeom g, const ChanJob *__restrict__ jobs, const uint32_t *__restrict__ chunk_tab,
                                                                uint32_t n_tiles, const cf32 *__restrict__ wtab_g, const cf32 *__restrict__ tw,
                                                                const cf32 *__restrict__ twc, float *__restrict__ minmax) {
    using B = BlockFft<14>;  // the half transforms: 16384 points
    constexpr int NH = B::NC, R = 16, NT = 1024;
    static_assert(B::T == NT && NH == R * SW::NC, "two 16384-point half transforms of sixteen wave transforms each");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    cf32 *const t2 = reinterpret_cast<cf32 *>(smem_raw);
    cf32 *const t3 = t2 + SW::T2_LEN;
    float *const red = reinterpret_cast<float *>(t3 + SW::T3_LEN);
    cf32 *const slabs = reinterpret_cast<cf32 *>(red + 2 * R);

    const uint32_t t = threadIdx.x, lane_w = t & 63u;
    const uint32_t wv = __builtin_amdgcn_readfirstlane(t >> 6);
    if (blockIdx.x >= n_tiles) return;
    // sub-transform tables: W_2048^i = tw[32 i] (tw = W_65536^i)
    constexpr uint32_t TS = 32;
    for (uint32_t i = t; i < (uint32_t)SW::T2_LEN; i += (uint32_t)NT) t2[i] = tw[TS * SW::t2_index(i / SW::NS2, i % SW::NS2)];
    for (uint32_t i = t; i < (uint32_t)SW::T3_LEN; i += (uint32_t)NT) {
        const uint32_t r = i / SW::NS3 + 1, k = i % SW::NS3;
        t3[i] = tw[TS * ((r * k) * (2u * SW::NC / (SW::NS3 * SW::R3)))];
    }
    const cf32 stw_a = tw[2 * t], stw_b = tw[2 * t + 1];  // W_65536^(2 t), W_65536^(2 t + 1): split twiddles of the even / odd bins; stw_a also = W_32768^t
    __syncthreads();
    cf32 *const slab = slabs + (size_t)wv * SUB_STRIDE;
    // exp(-i pi j / 16), j = 0 .. 15
    constexpr float RC[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f,
                              0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f, 0.0f, -0.19509032201612826785f,
                              -0.38268343236508977173f, -0.55557023301960222474f, -0.70710678118654752440f, -0.83146961230254523708f,
                              -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float RS[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f,
                              0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f, 1.0f, 0.98078528040323044913f,
                              0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f, 0.55557023301960222474f,
                              0.38268343236508977173f, 0.19509032201612826785f};
    float lmin = __builtin_inff(), lmax = -__builtin_inff();
    for (uint32_t ct = blockIdx.x; ct < n_tiles; ct += gridDim.x) {  // persistent: every gridDim.x-th chunk
        const uint32_t chan = chunk_tab[2 * (size_t)ct], f0 = chunk_tab[2 * (size_t)ct + 1];
        const uint32_t f1 = min(f0 + g.frames_per_tile, jobs[chan].f_end), spec_pitch = jobs[chan].spec_pitch;
        const gptr<const float> wav = as_global(jobs[chan].wav);
        const gptr<float> spec = as_global(jobs[chan].spec);
        for (uint32_t f = f0; f < f1; f++) {
            uint32_t tt = t, lane = lane_w;
            asm volatile("" : "+v"(tt), "+v"(lane));
            tt &= 1023u;
            lane &= 63u;
            const int64_t e0 = (int64_t)f * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;  // interior frames only
            // (opaque per-frame copies: the 32 rotated twiddles derived from them are loop-invariant, and hoisted out of the frame
            // loop they are 64 registers that get spilled — recomputing them is two to four operations each)
            cf32 sa = stw_a, sb = stw_b;
            asm volatile("" : "+v"(sa.re), "+v"(sa.im), "+v"(sb.re), "+v"(sb.im));
            cf32 ha[16];  // A[tt + 1024 c] (kept through the second half transform)
            cf32 z[16];
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                // ---- staging: thread tt forms a / b at the points tt + 1024 j (-> sub-transform tt mod 16, column tt / 16)
                {
                    // (an opaque thread index per pass: otherwise the second pass's loads are recognised as the first's and all 64
                    // loaded pairs are carried from one pass to the other — through scratch)
                    uint32_t tp = tt;
                    asm volatile("" : "+v"(tp));
                    tp &= 1023u;
                    const uint32_t c = tp >> 4, n1 = tp & 15u;
                    cf32 *const dst = slabs + (size_t)n1 * SUB_STRIDE + (((c >> 2) + 16u * (c & 3u) - n1) & 63u);
                    // four points at a time, the next batch's 16 loads in flight while this one is multiplied and stored (two sets of
                    // 32 registers; all 64 loads of a stage at once would be 128)
                    cf32 xl[2][4], xh[2][4], wl[2][4], wh[2][4];
                    auto request = [&](int h, int s_) {
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int j = 4 * h + i;
                            const gptr<const float> pl = wav + (e0 + 2 * NT * (int64_t)j), ph = pl + 2 * NH;
                            xl[s_][i] = {pl[2u * tp], pl[2u * tp + 1u]};
                            xh[s_][i] = {ph[2u * tp], ph[2u * tp + 1u]};
                            wl[s_][i] = (wtab_g + NT * j)[tp];
                            wh[s_][i] = (wtab_g + NH + NT * j)[tp];
                        }
                    };
                    // (the second pass keeps the first half spectrum in 32 registers: one batch in flight there)
                    if (pass == 0) request(0, 0);
#pragma unroll
                    for (int h = 0; h < 4; h++) {
                        if (pass == 0) {
                            if (h + 1 < 4) request(h + 1, (h + 1) & 1);
                        } else {
                            request(h, 0);
                        }
#pragma unroll
                        for (int i = 0; i < 4; i++) {
                            const int j = 4 * h + i, s_ = pass == 0 ? (h & 1) : 0;
                            const cf32 lo = {xl[s_][i].re * wl[s_][i].re, xl[s_][i].im * wl[s_][i].im}, hi = {xh[s_][i].re * wh[s_][i].re, xh[s_][i].im * wh[s_][i].im};
                            cf32 v;
                            if (pass == 0) {
                                v = {lo.re + hi.re, lo.im + hi.im};
                            } else {  // (lo - hi) W_32768^(tt + 1024 j) = (lo - hi) stw_a exp(-i pi j / 16)
                                const cf32 d = {lo.re - hi.re, lo.im - hi.im};
                                const cf32 w = cmul_c(sa, RC[j], -RS[j]);
                                v = cmul_c(d, w.re, w.im);
                            }
                            lds_st(&dst[64 * j], v);
                        }
                        __builtin_amdgcn_sched_barrier(0);  // (batch by batch: scheduled together, the 64 loads of a stage need 128 registers)
                    }
                }
                lds_barrier();
                // ---- the wave's 1024-point transform
                {
                    const uint32_t at = (lane - wv) & 63u;
#pragma unroll
                    for (int m = 0; m < 16; m++) z[m] = lds_ld(&slab[64 * m + at]);
                }
                wave_lds_sync_l();
                {
                    cf32 za[SW::NQ][SW::R3], zb[SW::NQ][SW::R3];
                    cf32 w2[SW::NT2];
                    SW::load_t2(lane, w2, t2);
                    SW::pass1(lane, z, slab);
                    wave_lds_sync_l();
                    SW::read1(lane, z, slab);
                    wave_lds_sync_l();
                    SW::pass2_w(lane, z, w2, slab);
                    wave_lds_sync_l();
                    cf32 wa[SW::NQ][SW::NT3], wb[SW::NQ][SW::NT3];
                    const typename SW::PairBase pbs = SW::pair_base(lane);
                    SW::load_t3_paired(pbs, wa, wb, t3);
                    SW::read2_paired(lane, pbs, za, zb, slab);
                    wave_lds_sync_l();
                    SW::pass3_paired_w(za, zb, wa, wb);
#pragma unroll
                    for (int q = 0; q < SW::NQ; q++) {
                        const uint32_t a = SW::jj_a(lane, q), b = SW::jj_b(lane, q);
#pragma unroll
                        for (int r = 0; r < SW::R3; r++) {
                            lds_st(&slab[a + (uint32_t)(r * SW::NS3)], za[q][r]);
                            lds_st(&slab[b + (uint32_t)(r * SW::NS3)], zb[q][r]);
                        }
                    }
                }
                // (the combining pass's constants: requested behind the sub-transform — its registers are free — they land during the barrier)
                cf32 wC[B::NTW];
#pragma unroll
                for (int e = 0; e < B::NTW; e++) wC[e] = (twc + NT * e)[tt];
                lds_barrier();
                // ---- combining pass: z[c] = half-spectrum bin tt + 1024 c
#pragma unroll
                for (int n1 = 0; n1 < 16; n1++) z[n1] = lds_ld(&slabs[(size_t)n1 * SUB_STRIDE + tt]);
                B::pass_last(z, wC);
                if (pass == 0) {
#pragma unroll
                    for (int c = 0; c < 16; c++) ha[c] = z[c];
                }
                lds_barrier();  // the slabs have been read: the next staging / the mirror exchange may write them
            }
            // ---- even bins: A through the n_fft 32768 plan's mirror exchange and split pass; results kept (bins 2 k, N - 2 k)
            const gptr<float> row = spec + (size_t)f * spec_pitch;
            float ra[17];
            {
                B::write_z(tt, ha, slabs);
                lds_barrier();
                cf32 zm[8];
                B::split_read(tt, slabs, zm);
                lds_barrier();
                // (BlockFft::split_compute's arithmetic, results into registers: bins k = tt + 1024 c -> ra[2 c], 16384 - k -> ra[2 c + 1])
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    const cf32 w = cmul_c(sa, RC[c], -RS[c]);  // W_32768^(tt + 1024 c)
                    const cf32 zk = ha[c];
                    const float er = zk.re + zm[c].re, ei = zk.im - zm[c].im;
                    const float dr = zk.re - zm[c].re, di = zk.im + zm[c].im;
                    const float xr = th_fma(di, w.re, th_fma(dr, w.im, er)), xi = th_fma(di, w.im, th_fma(-dr, w.re, ei));
                    const float yr = th_fma(2.0f, er, -xr), yi = th_fma(2.0f, ei, -xi);
                    const float px = xr * xr + xi * xi, py = yr * yr + yi * yi;
                    ra[2 * c] = AMP ? power_to_amp(px) : power_to_dB(px);
                    ra[2 * c + 1] = AMP ? power_to_amp(py) : power_to_dB(py);
                }
                {   // bin 8192 = Z[8 T] of thread 0: its own mirror, W^(Nc/2) = -i (only thread 0's value is used)
                    const cf32 zh = ha[8];
                    const float xr = 2.0f * zh.re, xi = -2.0f * zh.im;
                    const float p = xr * xr + xi * xi;
                    ra[16] = AMP ? power_to_amp(p) : power_to_dB(p);
                }
            }
            // ---- odd bins: B[k] pairs with B[16383 - k]
#pragma unroll
            for (int c = 0; c < 16; c++) lds_st(&slabs[tt + (uint32_t)(NT * c)], z[c]);
            lds_barrier();
            cf32 zm[8];
#pragma unroll
            for (int c = 0; c < 8; c++) zm[c] = lds_ld(&slabs[(uint32_t)(NH - 1) - tt - (uint32_t)(NT * c)]);
            lds_barrier();  // the slabs are free for the next frame's staging
            float fmn = __builtin_inff(), fmx = -__builtin_inff();
#pragma unroll
            for (int c = 0; c < 8; c++) {
                const cf32 w = cmul_c(sb, RC[c], -RS[c]);  // W_65536^(2 (tt + 1024 c) + 1)
                const cf32 zk = z[c];
                const float er = zk.re + zm[c].re, ei = zk.im - zm[c].im;
                const float dr = zk.re - zm[c].re, di = zk.im + zm[c].im;
                const float xr = th_fma(di, w.re, th_fma(dr, w.im, er)), xi = th_fma(di, w.im, th_fma(-dr, w.re, ei));
                const float yr = th_fma(2.0f, er, -xr), yi = th_fma(2.0f, ei, -xi);
                const float px = xr * xr + xi * xi, py = yr * yr + yi * yi;
                const float ox = AMP ? power_to_amp(px) : power_to_dB(px), oy = AMP ? power_to_amp(py) : power_to_dB(py);
                const uint32_t k = tt + (uint32_t)(NT * c);
                // (2 k, 2 k + 1) = (A's bin k, this x);  (N - 2 k - 1, N - 2 k) = (this y, A's bin 16384 - k)
                TH_SUBW2_STORE(&row[2u * k], ra[2 * c]);
                TH_SUBW2_STORE(&row[2u * k + 1u], ox);
                TH_SUBW2_STORE(&row[2u * (uint32_t)NH - 2u * k - 1u], oy);
                TH_SUBW2_STORE(&row[2u * (uint32_t)NH - 2u * k], ra[2 * c + 1]);
                if constexpr (!AMP) {
                    fmn = nmin_l(nmin_l(fmn, ox), nmin_l(oy, nmin_l(ra[2 * c], ra[2 * c + 1])));
                    fmx = nmax_l(nmax_l(fmx, ox), nmax_l(oy, nmax_l(ra[2 * c], ra[2 * c + 1])));
                }
            }
            if (tt == 0) {  // A's self-mirrored bin 8192 = bin 16384 of the frame
                row[(uint32_t)NH] = ra[16];
                if constexpr (!AMP) {
                    fmn = nmin_l(fmn, ra[16]);
                    fmx = nmax_l(fmx, ra[16]);
                }
            }
            lmin = nmin_l(lmin, fmn);
            lmax = nmax_l(lmax, fmx);
            {   // complete the row's last 128-byte line (see wave_frame)
                const uint32_t height = (uint32_t)(2 * NH + 1), padn = spec_pitch - height;
                if (tt - 1u < ((padn < 32u && spec_pitch % 32u == 0) ? padn : 0u)) row[height - 1u + tt] = 0.0f;
            }
        }
        if (minmax != nullptr) {  // the chunk's (min, max) pair
            const float a = wave_min_l(lmin), b = wave_max_l(lmax);
            if (lane_w == 0) {
                red[2 * wv] = a;
                red[2 * wv + 1] = b;
            }
            lds_barrier();
            if (t == 0) {
                float mn = red[0], mx = red[1];
                for (int w = 1; w < R; w++) {
                    mn = nmin_l(mn, red[2 * w]);
                    mx = nmax_l(mx, red[2 * w + 1]);
                }
                minmax[2 * (size_t)ct] = mn;
                minmax[2 * (size_t)ct + 1] = mx;
            }
            lmin = __builtin_inff();
            lmax = -__builtin_inff();
        }
    }
}

bool stft_subwave_applies(const StftGeom &g) { return g.log2_nc >= 12 && g.log2_nc <= 15 && g.odd_m1 == 0; }  // (15: stft_subwave2_kernel)
// Where it is the default (profiles/r05_ab_subwave_sizes.txt, ms for 128 ch x 30 s, subwave | block): n_fft 32768 1.26 | 1.85 (19200 / 4800: 2.40 |
// 3.26); 16384 at hop = n_fft / 4 0.98 | 1.00 (kept on the block kernel: its resident constants and two exchange buffers), 12000 / 3000 1.57 | 1.72;
// 8192 0.84 | 0.70, 3840 / 960 at 96 kHz 3.9 | 2.9 — four waves per workgroup are too few between two barriers, the block kernel stays.
bool stft_subwave_default(const StftGeom &g) {
    return stft_subwave_applies(g) && (g.log2_nc >= 14 || (g.log2_nc == 13 && g.hop * 4 != g.n_fft));
}
size_t stft_subwave_twc_len(const StftGeom &g) {
    return g.log2_nc >= 14 ? (size_t)SubwaveCfg<4>::NTWC * SubwaveCfg<4>::NT : g.log2_nc == 13 ? (size_t)SubwaveCfg<3>::NTWC * SubwaveCfg<3>::NT
                                                                                                 : (size_t)SubwaveCfg<2>::NTWC * SubwaveCfg<2>::NT;
}

// the combining pass's constants per thread, [constant e][thread t], from the host copy of tw (tw[i] = W_{n_fft}^i):
// R = 16: BlockFft<14>::load_tw<1024>;  R = 8: bins k2 = t + 512 u, (w, w^2, w^4, w W8) with w = W_Nc^k2 = tw[2 k2];  R = 4: k2 = t + 256 u, (w, w^2)
void stft_subwave_build_twc(const StftGeom &g, const cf32 *h_tw, cf32 *out) {
    const uint32_t n_fft = g.n_fft;
    if (g.log2_nc >= 14) {
        // (n_fft 65536: the half transforms are 16384-point ones — their twiddles are every second entry of the plan's table)
        using B = BlockFft<14>;
        std::vector<cf32> half;
        if (g.log2_nc == 15) {
            half.resize(n_fft / 2);
            for (uint32_t i = 0; i < n_fft / 2; i++) half[i] = h_tw[2 * i];
            h_tw = half.data();
        }
        for (uint32_t t = 0; t < 1024; t++) {
            cf32 w[B::NTW];
            B::template load_tw<B::NS_C>(t, w, h_tw);
            for (int e = 0; e < B::NTW; e++) out[(size_t)e * 1024 + t] = w[e];
        }
    } else if (g.log2_nc == 13) {
        for (uint32_t t = 0; t < 512; t++)
            for (uint32_t u = 0; u < 2; u++) {
                const uint32_t k2 = t + 512 * u;
                out[(size_t)(4 * u + 0) * 512 + t] = h_tw[(2 * k2) % n_fft];
                out[(size_t)(4 * u + 1) * 512 + t] = h_tw[(4 * k2) % n_fft];
                out[(size_t)(4 * u + 2) * 512 + t] = h_tw[(8 * k2) % n_fft];
                out[(size_t)(4 * u + 3) * 512 + t] = h_tw[(2 * k2 + n_fft / 8) % n_fft];
            }
    } else {
        for (uint32_t t = 0; t < 256; t++)
            for (uint32_t u = 0; u < 4; u++) {
                const uint32_t k2 = t + 256 * u;
                out[(size_t)(2 * u + 0) * 256 + t] = h_tw[(2 * k2) % n_fft];
                out[(size_t)(2 * u + 1) * 256 + t] = h_tw[(4 * k2) % n_fft];
            }
    }
}

namespace {
template <int LOG2_R>
hipError_t launch_subwave_r(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_chunk_tab, uint32_t n_tiles, const cf32 *d_wtab,
                            const cf32 *d_tw, const cf32 *d_twc, float *d_minmax, bool amp, uint32_t n_cu, hipStream_t s) {
    using C = SubwaveCfg<LOG2_R>;
    static_assert(C::LDS * C::WG_PER_CU + 64 <= 160 * 1024, "tables and slabs of the CU's workgroups fit its LDS");
    const bool reuse = TH_SUBW_REUSE && g.hop * 4 == g.n_fft;
    auto kern = amp ? (reuse ? stft_subwave_kernel<LOG2_R, true, true> : stft_subwave_kernel<LOG2_R, true, false>)
                    : (reuse ? stft_subwave_kernel<LOG2_R, false, true> : stft_subwave_kernel<LOG2_R, false, false>);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
    if (e != hipSuccess) return e;
    const uint32_t slots = n_cu * (uint32_t)C::WG_PER_CU;
    const uint32_t grid = n_tiles < slots ? n_tiles : slots;  // persistent: every workgroup walks every grid-th chunk
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS, s, g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_twc, d_minmax);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_stft_subwave(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_chunk_tab, uint32_t n_tiles, const cf32 *d_wtab,
                               const cf32 *d_tw, const cf32 *d_twc, float *d_minmax, bool amp, uint32_t n_cu, hipStream_t s) {
    if (!stft_subwave_applies(g) || n_cu == 0 || d_twc == nullptr) return hipErrorInvalidValue;
    if (!n_tiles) return hipSuccess;
    if (g.log2_nc == 15) {
        using C = SubwaveCfg<4>;
        auto kern = amp ? stft_subwave2_kernel<true> : stft_subwave2_kernel<false>;
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS);
        if (e != hipSuccess) return e;
        const uint32_t grid = n_tiles < n_cu ? n_tiles : n_cu;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), C::LDS, s, g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_twc, d_minmax);
        return hipGetLastError();
    }
    switch (g.log2_nc) {
        case 14: return launch_subwave_r<4>(g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_twc, d_minmax, amp, n_cu, s);
        case 13: return launch_subwave_r<3>(g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_twc, d_minmax, amp, n_cu, s);
#if TH_AB_VARIANTS  // (n_fft 8192 runs the block kernel: four waves between two barriers are too few; selector 15 of A/B builds, kernels.h)
        default: return launch_subwave_r<2>(g, d_jobs, d_chunk_tab, n_tiles, d_wtab, d_tw, d_twc, d_minmax, amp, n_cu, s);
#else
        default: return hipErrorInvalidValue;
#endif
    }
}

}  // namespace th
