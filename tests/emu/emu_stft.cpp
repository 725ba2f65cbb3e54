// emu_stft.cpp — CPU lane emulator of the wave STFT kernel (TEST SCAFFOLDING, never shipped):
// compiles thesia_amd/csrc/stft_wave.h with g++ and walks the 64 lanes sequentially through the
// same phase functions the gfx950 kernel runs, so the index arithmetic (Stockham passes, LDS
// swizzle, mirror exchange, split pass) can be checked against the oracle without a GPU.
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../thesia_amd/csrc/stft_wave.h"
#include "../../thesia_amd/csrc/stft_wave_multi.h"
#include <type_traits>

#include "../../thesia_amd/csrc/stft_block.h"
#include "../../thesia_amd/csrc/mel_fuse.h"
#include "../../thesia_amd/csrc/host_math.h"  // bluestein_tables (host_math.cpp is compiled into the emulator)

using namespace th;

template <int LOG2_NC>
static void emu_frame(const float *wav, uint32_t n_samples, uint32_t frame, const StftGeom &g, const cf32 *wtab,
                      const cf32 *tw, float *row) {
    using W = WaveFft<LOG2_NC>;
    constexpr int P = W::P, NC = W::NC;
    std::vector<cf32> slab(W::SLAB_LEN);
    static cf32 x[64][P], z[64][P];
    std::vector<cf32> t2(W::T2_LEN), t3(W::T3_LEN);
    const int64_t e0 = (int64_t)frame * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
    for (uint32_t t = 0; t < 256; t++) W::fill_tables(t, 256, tw, t2.data(), t3.data());
    // (a lane's column of the frame is W::lane_col(lane): the plane exchanges of n_fft = 2048 deal the columns out permuted)
    for (uint32_t l = 0; l < 64; l++) wave_fetch<P, 0>(W::lane_col(l), x[l], wav, e0);
    for (uint32_t l = 0; l < 64; l++) wave_window<P>(W::lane_col(l), z[l], x[l], wtab);
    for (uint32_t l = 0; l < 64; l++) W::pass1(l, z[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) W::read1(l, z[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) W::pass2(l, z[l], t2.data(), slab.data());
    if (W::PAIRED) {  // mirror-local last pass: no third exchange
        static cf32 za[64][W::NQ][W::R3], zb[64][W::NQ][W::R3];
        for (uint32_t l = 0; l < 64; l++) W::read2_paired(l, za[l], zb[l], slab.data());
        for (uint32_t l = 0; l < 64; l++) W::pass3_paired(l, za[l], zb[l], t3.data());
        std::vector<int> hits(NC + 1, 0);
        for (uint32_t l = 0; l < 64; l++)
            W::split_paired(l, za[l], zb[l], tw, [&](uint32_t kb, int kc, float p) {
                const uint32_t k = kb + (uint32_t)kc;
                row[k] = power_to_dB(p);
                hits[k]++;
            });
        for (int k = 0; k <= NC; k++)
            if (hits[k] != 1) row[k] = NAN;  // every bin must be emitted exactly once
        return;
    }
    if constexpr (W::SWAP8) {  // n_fft 1024: the wave's halves swap registers for the split pass (v_permlane32_swap on the GPU)
        static cf32 z256[64];
        typename W::Swap8Lane sl[64];
        for (uint32_t l = 0; l < 64; l++) sl[l] = W::swap8_lane(l);
        for (uint32_t l = 0; l < 64; l++) W::read2_sw(sl[l], z[l], slab.data());
        for (uint32_t l = 0; l < 64; l++) {
            cf32 w3[4];
            W::load_t3_sw(sl[l], w3, t3.data());
            W::pass3_sw(sl[l], z[l], w3, z256[l]);
        }
        // swap(a = z[i], b = z[7 - i]): lanes 32..63 of a <-> lanes 0..31 of b, lanes 0 and 32 masked out
        for (uint32_t l = 1; l < 32; l++)
            for (int i = 0; i < 4; i++) std::swap(z[32 + l][i % P], z[l][(7 - i) % P]);
        std::vector<int> hits(NC + 1, 0);
        for (uint32_t l = 0; l < 64; l++)
            W::split_sw(sl[l], z[l], z256[l], tw, [&](uint32_t kb, int kc, float p) {
                const uint32_t k = kb + (uint32_t)kc;
                row[k] = power_to_dB(p);
                hits[k]++;
            });
        for (int k = 0; k <= NC; k++)
            if (hits[k] != 1) row[k] = NAN;  // every bin must be emitted exactly once
    }
}

// The packed-f32 pipeline of the n_fft 2048 plan (WaveFft<10>::*_pk, stft_pk.h): the same phases on register pairs, with
// the pair-ordered split-twiddle table.  On the CPU the pair helpers are plain structs: this checks the pair bookkeeping
// (which point sits in which half of which pair after every level), the lane-0 selects and the table layout.
static void emu_frame_pk(const float *wav, uint32_t n_samples, uint32_t frame, const StftGeom &g, const cf32 *wtab,
                         const cf32 *tw, float *row) {
    using W = WaveFft<10>;
    constexpr int P = W::P, NC = W::NC;
    static_assert(W::PK, "packed pipeline");
    std::vector<cf32> slab(W::SLAB_LEN), t2(W::T2_LEN), t3(W::T3_LEN), stwp(W::STWP_LEN);
    static cf32 x[64][P], z[64][P];
    static v2f zp[64][16], zr[64][8], zi[64][8];
    const int64_t e0 = (int64_t)frame * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
    for (uint32_t t = 0; t < 256; t++) {
        W::fill_tables(t, 256, tw, t2.data(), t3.data());
        W::fill_stwp(t, 256, tw, stwp.data());
    }
    for (uint32_t l = 0; l < 64; l++) wave_fetch<P, 0>(W::lane_col(l), x[l], wav, e0);
    for (uint32_t l = 0; l < 64; l++) wave_window<P>(W::lane_col(l), z[l], x[l], wtab);
    for (uint32_t l = 0; l < 64; l++)
        for (int m = 0; m < P; m++) zp[l][m] = mk2(z[l][m].re, z[l][m].im);
    for (uint32_t l = 0; l < 64; l++) W::pass1_pk(l, zp[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) W::read1_pk(l, zr[l], zi[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) {
        v2f w2[W::NT2];
        W::load_t2_pk(l, w2, t2.data());
        W::pass2_pk(l, zr[l], zi[l], w2, slab.data());
    }
    static typename W::PkPairs za[64][W::NQ], zb[64][W::NQ];
    for (uint32_t l = 0; l < 64; l++) W::read2_paired_pk(W::pair_base(l), za[l], zb[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) {
        v2f wa[W::NQ][W::NT3], wb[W::NQ][W::NT3];
        W::load_t3_paired_pk(W::pair_base(l), wa, wb, t3.data());
        W::pass3_paired_pk(za[l], zb[l], wa, wb);
    }
    std::vector<int> hits(NC + 1, 0);
    for (uint32_t l = 0; l < 64; l++) {
        v2f wsr[W::NQ][2], wsi[W::NQ][2];
        W::load_stw_paired_pk(l, wsr, wsi, stwp.data());
        W::split_paired_pk(l, za[l], zb[l], wsr, wsi, tw[NC / 2], [&](uint32_t kb, int kc, float p) {
            const uint32_t k = kb + (uint32_t)kc;
            row[k] = power_to_dB(p);
            hits[k]++;
        });
    }
    for (int k = 0; k <= NC; k++)
        if (hits[k] != 1) row[k] = NAN;  // every bin must be emitted exactly once
}

// One frame of the workgroup-per-frame plan (stft_block.h): the T threads run every phase one after the other, a phase
// boundary stands for the workgroup barrier.
template <int LOG2_NC>
static void emu_frame_block(const float *wav, uint32_t frame, const StftGeom &g, const cf32 *wtab, const cf32 *tw, float *row) {
    using B = BlockFft<LOG2_NC>;
    constexpr int T = B::T, NC = B::NC;
    std::vector<cf32> buf(B::BUF_LEN);
    static cf32 z[T][16];
    const int64_t e0 = (int64_t)frame * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
    for (uint32_t t = 0; t < (uint32_t)T; t++)
        for (int m = 0; m < 16; m++) {
            const uint32_t n = t + (uint32_t)T * m;
            z[t][m] = {wav[e0 + 2 * n] * wtab[n].re, wav[e0 + 2 * n + 1] * wtab[n].im};
        }
    // an exchange: every thread stores, (barrier), every thread reads; planar plans (n_fft 65536) run it per part
    auto exchange = [&](auto store, auto read) {
        if constexpr (B::PLANAR) {
            for (uint32_t t = 0; t < (uint32_t)T; t++) store(t, std::integral_constant<int, 0>{});
            for (uint32_t t = 0; t < (uint32_t)T; t++) read(t, std::integral_constant<int, 0>{});
            for (uint32_t t = 0; t < (uint32_t)T; t++) store(t, std::integral_constant<int, 1>{});
            for (uint32_t t = 0; t < (uint32_t)T; t++) read(t, std::integral_constant<int, 1>{});
        } else {
            for (uint32_t t = 0; t < (uint32_t)T; t++) store(t, std::integral_constant<int, -1>{});
            for (uint32_t t = 0; t < (uint32_t)T; t++) read(t, std::integral_constant<int, -1>{});
        }
    };
    for (uint32_t t = 0; t < (uint32_t)T; t++) B::pass_first_compute(z[t]);
    exchange([&](uint32_t t, auto part) { B::template pass_first_store<decltype(part)::value>(t, z[t], buf.data()); },
             [&](uint32_t t, auto part) { B::template read_in<B::FIRST_LAYOUT, decltype(part)::value>(t, z[t], buf.data()); });
    if (B::R2_FIRST) {
        for (uint32_t t = 0; t < (uint32_t)T; t++) {
            cf32 w[B::NTW];
            B::template load_tw<B::NS_A>(t, w, tw);
            B::template pass_mid_compute<B::NS_A>(z[t], w);
        }
        exchange([&](uint32_t t, auto part) { B::template pass_mid_store<B::NS_A, decltype(part)::value>(t, z[t], buf.data()); },
                 [&](uint32_t t, auto part) { B::template read_in<B::NS_A, decltype(part)::value>(t, z[t], buf.data()); });
    }
    for (uint32_t t = 0; t < (uint32_t)T; t++) {
        cf32 w[B::NTW];
        B::template load_tw<B::NS_B>(t, w, tw);
        B::template pass_mid_compute<B::NS_B>(z[t], w);
    }
    exchange([&](uint32_t t, auto part) { B::template pass_mid_store<B::NS_B, decltype(part)::value>(t, z[t], buf.data()); },
             [&](uint32_t t, auto part) { B::template read_in<B::NS_B, decltype(part)::value>(t, z[t], buf.data()); });
    for (uint32_t t = 0; t < (uint32_t)T; t++) {
        cf32 w[B::NTW];
        B::template load_tw<B::NS_C>(t, w, tw);
        B::pass_last(z[t], w);
    }
    static cf32 zm[T][8];
    exchange([&](uint32_t t, auto part) { B::template write_z<decltype(part)::value>(t, z[t], buf.data()); },
             [&](uint32_t t, auto part) { B::template split_read<decltype(part)::value>(t, buf.data(), zm[t]); });
    std::vector<int> hits(NC + 1, 0);
    for (uint32_t t = 0; t < (uint32_t)T; t++)
        B::split_compute(t, z[t], zm[t], tw[t], [&](uint32_t k, float p) {
            row[k] = power_to_dB(p);
            hits[k]++;
        });
    for (int k = 0; k <= NC; k++)
        if (hits[k] != 1) row[k] = NAN;  // every bin exactly once
}

// G frames of one wave of the multi-frame plan (stft_wave_multi.h): group g of the lanes computes frame frames[g]
template <int LOG2_NC>
static void emu_frames_multi(const float *wav, const uint32_t *frames, const StftGeom &g, const cf32 *wtab, const cf32 *tw,
                             float *const *rows) {
    using W = WaveFftM<LOG2_NC>;
    constexpr int P = W::P, NC = W::NC;
    std::vector<cf32> slab(W::SLAB_LEN);
    static cf32 x[64][P], z[64][P];
    std::vector<cf32> t2(W::T2_LEN), t3(W::T3_LEN + 1);
    for (uint32_t t = 0; t < 256; t++) W::fill_tables(t, 256, tw, t2.data(), t3.data());
    for (uint32_t l = 0; l < 64; l++) {
        const int64_t e0 = (int64_t)frames[W::grp(l)] * g.hop - (int64_t)(g.win / 2) - (int64_t)g.pad_left;
        // slot m of a lane = complex point col + L m of its frame
        for (int m = 0; m < P; m++) {
            const uint32_t n = W::lane_col(l) + (uint32_t)W::L * m;
            x[l][m] = {wav[e0 + 2 * n], wav[e0 + 2 * n + 1]};
            const cf32 w = wtab[n];
            z[l][m] = {x[l][m].re * w.re, x[l][m].im * w.im};
        }
    }
    for (uint32_t l = 0; l < 64; l++) W::pass1(l, z[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) W::read1(l, z[l], slab.data());
    {
        static cf32 w2[64][W::NT2];
        for (uint32_t l = 0; l < 64; l++) W::load_t2(l, w2[l], t2.data());
        for (uint32_t l = 0; l < 64; l++) W::pass2_w(l, z[l], w2[l], slab.data());
    }
    static cf32 za[64][W::NQ][W::R3], zb[64][W::NQ][W::R3];
    for (uint32_t l = 0; l < 64; l++) W::read2_paired(l, za[l], zb[l], slab.data());
    for (uint32_t l = 0; l < 64; l++) {
        cf32 wa[W::NW3], wb[W::NW3];
        W::load_t3_paired(l, wa, wb, t3.data());
        W::pass3_paired_w(za[l], zb[l], wa, wb);
    }
    std::vector<int> hits((size_t)W::G * (NC + 1), 0);
    for (uint32_t l = 0; l < 64; l++) {
        cf32 ws[W::NQ][W::R3];
        W::load_stw_paired(l, ws, tw);
        const uint32_t gi = W::grp(l);
        W::split_paired_w(l, za[l], zb[l], ws, tw[NC / 2], [&](uint32_t kb, int kc, float p) {
            const uint32_t k = kb + (uint32_t)kc;
            rows[gi][k] = power_to_dB(p);
            hits[gi * (NC + 1) + k]++;
        });
    }
    for (int gi = 0; gi < W::G; gi++)
        for (int k = 0; k <= NC; k++)
            if (hits[gi * (NC + 1) + k] != 1) rows[gi][k] = NAN;  // every bin exactly once per frame
}

// the multi-frame plan (n_fft = 512, 1024): out as emu_stft_wave
extern "C" __attribute__((visibility("default"))) int emu_stft_wave_multi(const float *wav, uint32_t n_samples, uint32_t win,
                                                                           uint32_t hop, uint32_t n_fft, const float *window,
                                                                           uint32_t n_frames, float *out) {
    if (n_fft != 512 && n_fft != 1024) return -1;
    StftGeom g{};
    g.hop = hop;
    g.win = win;
    g.n_fft = n_fft;
    g.pad_left = (n_fft - win) / 2;
    g.nc = n_fft / 2;
    g.n_freq = n_fft / 2 + 1;
    g.height = g.n_freq;
    std::vector<cf32> tw(n_fft), wtab(g.nc);
    for (uint32_t i = 0; i < n_fft; i++) {
        const double a = -2.0 * M_PI * (double)i / (double)n_fft;
        tw[i] = {(float)std::cos(a), (float)std::sin(a)};
    }
    std::vector<float> wpad(n_fft, 0.0f);
    for (uint32_t i = 0; i < win; i++) wpad[g.pad_left + i] = 0.5f * WAVE_PRESCALE * window[i];
    for (uint32_t n = 0; n < g.nc; n++) wtab[n] = {wpad[2 * n], wpad[2 * n + 1]};
    std::vector<uint32_t> interior;
    for (uint32_t f = 0; f < n_frames; f++) {
        const int64_t e0 = (int64_t)f * hop - (int64_t)(win / 2) - (int64_t)g.pad_left;
        if (e0 < 0 || e0 + (int64_t)n_fft > (int64_t)n_samples)
            for (uint32_t k = 0; k < g.n_freq; k++) out[(size_t)f * g.n_freq + k] = NAN;
        else interior.push_back(f);
    }
    const uint32_t G = 1024 / g.nc;
    std::vector<float> scratch(g.n_freq);
    for (size_t i = 0; i < interior.size(); i += G) {
        uint32_t fr[4];
        float *rows[4];
        for (uint32_t k = 0; k < G; k++) {
            const bool valid = i + k < interior.size();
            fr[k] = interior[valid ? i + k : interior.size() - 1];  // the kernel clamps: a duplicate frame, computed again
            rows[k] = valid ? out + (size_t)fr[k] * g.n_freq : scratch.data();
        }
        if (n_fft == 1024) emu_frames_multi<9>(wav, fr, g, wtab.data(), tw.data(), rows);
        else emu_frames_multi<8>(wav, fr, g, wtab.data(), tw.data(), rows);
    }
    return 0;
}

// out: n_frames x (n_fft/2+1).  window: normalised window (len win).  Returns 0 on success.
static int emu_stft_wave_impl(const float *wav, uint32_t n_samples, uint32_t win, uint32_t hop, uint32_t n_fft,
                              const float *window, uint32_t n_frames, float *out, bool packed);
extern "C" __attribute__((visibility("default"))) int emu_stft_wave(const float *wav, uint32_t n_samples,
                                                                     uint32_t win, uint32_t hop, uint32_t n_fft,
                                                                     const float *window, uint32_t n_frames,
                                                                     float *out) {
    return emu_stft_wave_impl(wav, n_samples, win, hop, n_fft, window, n_frames, out, false);
}
// the packed-f32 pipeline (n_fft 2048 only)
extern "C" __attribute__((visibility("default"))) int emu_stft_wave_pk(const float *wav, uint32_t n_samples,
                                                                        uint32_t win, uint32_t hop, uint32_t n_fft,
                                                                        const float *window, uint32_t n_frames,
                                                                        float *out) {
    if (n_fft != 2048) return -1;
    return emu_stft_wave_impl(wav, n_samples, win, hop, n_fft, window, n_frames, out, true);
}
static int emu_stft_wave_impl(const float *wav, uint32_t n_samples, uint32_t win, uint32_t hop, uint32_t n_fft,
                              const float *window, uint32_t n_frames, float *out, bool packed) {
    StftGeom g{};
    g.hop = hop;
    g.win = win;
    g.n_fft = n_fft;
    g.pad_left = (n_fft - win) / 2;
    g.nc = n_fft / 2;
    g.n_freq = n_fft / 2 + 1;
    g.height = g.n_freq;
    std::vector<cf32> tw(n_fft), wtab(g.nc);
    for (uint32_t i = 0; i < n_fft; i++) {
        const double a = -2.0 * M_PI * (double)i / (double)n_fft;
        tw[i] = {(float)std::cos(a), (float)std::sin(a)};
    }
    std::vector<float> wpad(n_fft, 0.0f);
    for (uint32_t i = 0; i < win; i++) wpad[g.pad_left + i] = 0.5f * WAVE_PRESCALE * window[i];  // as api.hip builds the table
    for (uint32_t n = 0; n < g.nc; n++) wtab[n] = {wpad[2 * n], wpad[2 * n + 1]};
    for (uint32_t f = 0; f < n_frames; f++) {
        float *row = out + (size_t)f * g.n_freq;
        const int64_t e0 = (int64_t)f * hop - (int64_t)(win / 2) - (int64_t)g.pad_left;
        if (e0 < 0 || e0 + (int64_t)n_fft > (int64_t)n_samples) {  // boundary frame: not the wave kernel's job
            for (uint32_t k = 0; k < g.n_freq; k++) row[k] = NAN;
            continue;
        }
        switch (n_fft) {
            case 1024: emu_frame<9>(wav, n_samples, f, g, wtab.data(), tw.data(), row); break;
            case 2048:
                if (packed) emu_frame_pk(wav, n_samples, f, g, wtab.data(), tw.data(), row);
                else emu_frame<10>(wav, n_samples, f, g, wtab.data(), tw.data(), row);
                break;
            case 4096: emu_frame<11>(wav, n_samples, f, g, wtab.data(), tw.data(), row); break;
            case 8192: emu_frame_block<12>(wav, f, g, wtab.data(), tw.data(), row); break;
            case 16384: emu_frame_block<13>(wav, f, g, wtab.data(), tw.data(), row); break;
            case 32768: emu_frame_block<14>(wav, f, g, wtab.data(), tw.data(), row); break;
            case 65536: emu_frame_block<15>(wav, f, g, wtab.data(), tw.data(), row); break;
            default: return -1;
        }
    }
    return 0;
}

// Check of the quotient the quantise kernel uses (kernels_image.hip: quantise): q0 = a*r, e = fma(-q0, b, a),
// q = fma(e, r, q0) with r = RN(1/b) against the IEEE quotient a / b.  Returns the number of mismatches (signed
// zeros compare equal).  fmaf is the correctly rounded C library / hardware operation.
extern "C" __attribute__((visibility("default"))) uint64_t emu_fma_div_mismatches(const float *a, uint64_t n, float b) {
    const float r = 1.0f / b;
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; i++) {
        const float want = a[i] / b;
        const float q0 = a[i] * r;
        const float e = std::fmaf(-q0, b, a[i]);
        const float q = std::fmaf(e, r, q0);
        if (std::memcmp(&q, &want, 4) != 0 && !(q == 0.0f && want == 0.0f)) bad++;
    }
    return bad;
}

// Fused mel epilogue (mel_fuse.h tables + stft_wave.h lane functions) on one amplitude row: out[m] = the filter outputs
// (linear, before dB).  info[0..2] = pieces, slots, groups.  Returns 0, or 1 when the filterbank is not fusable.
extern "C" __attribute__((visibility("default"))) int emu_mel_fuse(const float *amp, const float *fb, uint32_t n_freq,
                                                                    uint32_t n_mel, uint32_t max_pieces, float *out,
                                                                    uint32_t *info) {
    const MelFuseHost h = build_mel_fuse(fb, n_freq, n_mel, max_pieces);
    if (!h.ok) return 1;
    info[0] = h.n_pieces;
    info[1] = h.n_slots;
    info[2] = h.n_groups;
    if (h.words.size() != mel_fuse_words(h.n_slots, h.n_groups)) return -1;
    const MelFuseTab t = mel_fuse_view(h.words.data(), h.n_slots, h.n_groups);
    std::vector<cf32> prf(64 * (size_t)h.n_slots);
    for (uint32_t m = 0; m < n_mel; m++) out[m] = NAN;
    for (uint32_t l = 0; l < 64; l++) mel_pieces(l, amp, prf.data(), t);
    for (uint32_t l = 0; l < 64; l++)
        mel_gather(l, prf.data(), t, [&](uint32_t m, float v) {
            if (m < n_mel) out[m] = v;
        });
    return 0;
}

// Frame pairs (mel_banded_pair, round 5): two amplitude rows RB floats apart, one pass over the paired table; out_a / out_b[m] and
// (every lane's result, also those past the last mel) the extremes over all lanes in mm[0..3] = min a, max a, min b, max b — the
// kernel folds every lane into its min / max because lanes past the last mel repeat the last filter.  reach = MelBandHost::reach.
extern "C" __attribute__((visibility("default"))) int emu_mel_band_pair(const float *amp_a, const float *amp_b, const float *fb, uint32_t n_freq,
                                                                         uint32_t n_mel, float *out_a, float *out_b, float *mm, uint32_t *reach) {
    constexpr int RB = 1092;
    const MelBandHost h = build_mel_band(fb, n_freq, n_mel, 1u << 16, true, true);
    if (!h.ok) return 1;
    *reach = h.reach;
    if (n_freq > RB - 64 || h.reach > n_freq + 59) return 2;
    std::vector<float> a(2 * RB + 64, NAN);  // (NaN wherever the kernel guarantees nothing)
    for (uint32_t k = 0; k < n_freq; k++) {
        a[k] = amp_a[k];
        a[RB + k] = amp_b[k];
    }
    for (uint32_t k = 0; k < 64; k++) a[n_freq + k] = 0.0f;   // the zeros wave_frame writes behind the rows
    for (uint32_t k = 0; k < 59; k++) a[RB + n_freq + k] = 0.0f;
    uint32_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t g = 0; g < h.n_groups; g++) {
        off[g] = h.words[2 * g];
        nt[g] = h.words[2 * g + 1];
    }
    mm[0] = mm[2] = INFINITY;
    mm[1] = mm[3] = -INFINITY;
    for (uint32_t l = 0; l < 64; l++)
        mel_banded_pair<RB>(
            l, a.data(), h.words.data(), h.n_groups, off, nt,
            [&](uint32_t m, float v) {
                mm[0] = std::fmin(mm[0], v);
                mm[1] = std::fmax(mm[1], v);
                if (m < n_mel) out_a[m] = v;
            },
            [&](uint32_t m, float v) {
                mm[2] = std::fmin(mm[2], v);
                mm[3] = std::fmax(mm[3], v);
                if (m < n_mel) out_b[m] = v;
            });
    return 0;
}

// The same product as banded sums, lane = mel (build_mel_band + mel_banded): out[m] = the filter outputs (linear).
// layout: 0 = first bins as the filters start, 1 = spread over the LDS banks, 2 = the paired layout (even first bins, weights
// in quads).  info[0..4] = table words, groups, widest group's taps, all groups' taps, LDS cycles of one frame's amplitude reads
// by the bank rule (mel_fuse.h).  amp must be followed by 128 readable zeros (the kernel zeroes them).
extern "C" __attribute__((visibility("default"))) int emu_mel_band(const float *amp, const float *fb, uint32_t n_freq,
                                                                    uint32_t n_mel, uint32_t max_words, uint32_t layout, float *out,
                                                                    uint32_t *info) {
    const MelBandHost h = build_mel_band(fb, n_freq, n_mel, max_words, layout >= 1, layout == 2);
    if (!h.ok) return 1;
    info[0] = (uint32_t)h.words.size();
    info[1] = h.n_groups;
    info[2] = h.max_taps;
    info[3] = info[4] = 0;
    std::vector<float> a(n_freq + MEL_BAND_MAX_TAPS, 0.0f);
    std::memcpy(a.data(), amp, n_freq * sizeof(float));
    for (uint32_t m = 0; m < n_mel; m++) out[m] = NAN;
    uint32_t off[8] = {0, 0, 0, 0, 0, 0, 0, 0}, nt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (uint32_t g = 0; g < h.n_groups; g++) {
        off[g] = h.words[2 * g];
        nt[g] = h.words[2 * g + 1];
        info[3] += nt[g];
        uint32_t first[64];
        for (uint32_t l = 0; l < 64; l++) {
            first[l] = h.words[off[g] + l];
            if (layout == 2) {
                if (first[l] & 1u) return 2;  // 8-byte reads need even first bins
                first[l] /= 2;
            }
        }
        const uint32_t cyc = mel_band_half_cycles(first, 32) + mel_band_half_cycles(first + 32, 32);
        info[4] += layout == 2 ? nt[g] / 2 * cyc : nt[g] * cyc;
    }
    for (uint32_t l = 0; l < 64; l++) {
        auto emit = [&](uint32_t m, float v) {
            if (m < n_mel) out[m] = v;
        };
        if (layout == 2) mel_banded<true>(l, a.data(), h.words.data(), h.n_groups, off, nt, emit);
        else mel_banded<false>(l, a.data(), h.words.data(), h.n_groups, off, nt, emit);
    }
    return 0;
}

// Moment form of the mel filterbank (round 6: build_mel_moments + mel_mom_lane / mel_mom_w_lane / mel_mom_combine, lane = segment):
// out[m] = the filter outputs (linear).  amp_slab: the amplitude row followed by whatever the slab holds behind it (the caller
// passes NaN there: the masks must keep it out).  info[0..5] = table words, groups, all groups' taps, widest group's taps, reach,
// groups in the W form; dev[0..1] = MelMomHost::max_dev, max_amp.  Returns 0, 1 when the filterbank has no moment form.
extern "C" __attribute__((visibility("default"))) int emu_mel_moments(const float *amp_slab, uint32_t slab_len, const float *fb, const float *lin,
                                                                       const float *mf, uint32_t n_freq, uint32_t n_mel, uint32_t spread, float *out,
                                                                       uint32_t *info, double *dev) {
    const MelMomHost h = build_mel_moments(fb, lin, mf, n_freq, n_mel, slab_len, spread != 0);
    dev[0] = h.max_dev;
    dev[1] = h.max_amp;
    if (!h.ok) return 1;
    info[0] = (uint32_t)h.words.size();
    info[1] = h.n_groups;
    info[2] = h.taps;
    info[3] = h.max_taps;
    info[4] = h.reach;
    info[5] = h.w_groups;
    dev[0] = h.max_dev;
    dev[1] = h.max_amp;
    if (h.words[0] != h.n_groups || h.reach > slab_len) return -1;
    for (uint32_t m = 0; m < n_mel; m++) out[m] = NAN;
    float carry = 0.0f;  // F of lane 0 of the group above (the kernel walks the groups downwards)
    for (uint32_t g = h.n_groups; g-- != 0;) {
        const uint32_t nw = h.words[MEL_MOM_HDR0 + 2 * g], n = nw & 0xffffu, off = h.words[MEL_MOM_HDR0 + 1 + 2 * g];
        const bool wform = (nw & MEL_MOM_FORM_W) != 0;
        if (off % 4 != 0 || (wform ? (n < 1 || n > 2) : n % MEL_MOM_UNROLL != 0)) return -2;
        const uint64_t *masks = reinterpret_cast<const uint64_t *>(h.words.data() + off + 256);
        MelMomLane s[64];
        for (uint32_t l = 0; l < 64; l++) {
            float prm[3], w1[2];
            std::memcpy(prm, &h.words[off + 4 * l + 1], 12);
            std::memcpy(w1, &h.words[off + 256 + 2 * l], 8);
            s[l] = wform ? mel_mom_w_lane(amp_slab, h.words[off + 4 * l], n, prm[0], prm[1], w1[0], w1[1])
                         : mel_mom_lane_any(l, amp_slab, h.words[off + 4 * l], prm[0], prm[1], masks, n);
        }
        if (!wform) {  // the window words say what the masks say: the lockstep walk of the workgroup-per-frame kernels, bit for bit
            const uint32_t *win = h.words.data() + off + 256 + mel_mom_mask_words(n);
            uint32_t n_run = n;  // (the widest M group of the batch of MEL_MOM_BATCH groups this one is walked with)
            for (uint32_t gg = g / MEL_MOM_BATCH * MEL_MOM_BATCH; gg < std::min(h.n_groups, (g / MEL_MOM_BATCH + 1) * MEL_MOM_BATCH); gg++) {
                const uint32_t nw2 = h.words[MEL_MOM_HDR0 + 2 * gg];
                if (!(nw2 & MEL_MOM_FORM_W)) n_run = std::max(n_run, nw2 & 0xffffu);
            }
            for (uint32_t l = 0; l < 64; l++) {
                float prm[2];
                std::memcpy(prm, &h.words[off + 4 * l + 1], 8);
                const uint32_t first = h.words[off + 4 * l];
                // (the wave kernel's slab ends closer behind the row than a workgroup's exchange buffer)
                const MelMomLane w = mel_mom_lane_win(amp_slab, first, prm[0], prm[1], win[l], first + n_run <= slab_len ? n_run : n);
                if (std::memcmp(&w, &s[l], sizeof w) != 0) return -4;
            }
        }
        for (uint32_t l = 0; l < 64; l++) {
            float inv_d;
            std::memcpy(&inv_d, &h.words[off + 4 * l + 3], 4);
            const float v = mel_mom_combine(inv_d, s[l].R, l == 63 ? carry : s[l + 1].F);
            const uint32_t m = 64 * g + l;
            if (m < n_mel) out[m] = v;
            else if (v != 0.0f) return -3;  // lanes past the last mel hold 1 / d = 0
        }
        carry = s[0].F;
    }
    // The workgroup-per-frame kernels' lane table (made without spreading: a lane's window starts at its first bin): walked as
    // mel_moments_range_lockstep walks it — batches of MEL_MOM_BATCH groups counted from 0, every M group of a batch over the taps
    // of the widest — the same numbers bit for bit.
    if (spread == 0) {
        const std::vector<uint32_t> lt = build_mel_mom_lanes(h);
        if (lt.size() != MEL_LANE_BLK0 + (size_t)MEL_LANE_STRIDE * (h.n_groups + MEL_MOM_BATCH)) return -6;
        for (uint32_t i = 0; i < MEL_LANE_BLK0; i++)
            if (lt[i] != h.words[i]) return -6;
        float carry2 = 0.0f;
        for (uint32_t g = h.n_groups; g-- != 0;) {
            const uint32_t *blk = &lt[MEL_LANE_BLK0 + (size_t)MEL_LANE_STRIDE * g];
            const uint32_t nw = blk[256 + 2], n = nw & 0xffffu;
            uint32_t n_run = 0;
            for (uint32_t gg = g / MEL_MOM_BATCH * MEL_MOM_BATCH; gg < std::min(h.n_groups, (g / MEL_MOM_BATCH + 1) * MEL_MOM_BATCH); gg++) {
                const uint32_t nw2 = lt[MEL_LANE_BLK0 + (size_t)MEL_LANE_STRIDE * gg + 256 + 2];
                if (!(nw2 & MEL_MOM_FORM_W)) n_run = std::max(n_run, nw2 & 0xffffu);
            }
            MelMomLane s[64];
            for (uint32_t l = 0; l < 64; l++) {
                if (blk[256 + 4 * l + 2] != nw) return -7;
                float p[4], q[2];
                std::memcpy(p, &blk[4 * l], 16);
                std::memcpy(q, &blk[256 + 4 * l], 8);
                const uint32_t first = blk[4 * l], win = blk[256 + 4 * l];
                if (nw & MEL_MOM_FORM_W) {
                    s[l] = mel_mom_w_lane(amp_slab, first, n, p[1], p[2], q[0], q[1]);
                } else {
                    if ((win & 0xffffu) != 0) return -7;
                    s[l] = mel_mom_lane_win(amp_slab, first, p[1], p[2], win, first + n_run <= slab_len ? n_run : n);
                }
            }
            for (uint32_t l = 0; l < 64; l++) {
                float inv_d;
                std::memcpy(&inv_d, &blk[4 * l + 3], 4);
                const float v = mel_mom_combine(inv_d, s[l].R, l == 63 ? carry2 : s[l + 1].F);
                const uint32_t m = 64 * g + l;
                if (m < n_mel && std::memcmp(&v, &out[m], 4) != 0) return -8;
            }
            carry2 = s[0].F;
        }
        // the waves' ranges: ascending, from 0 to the last group
        if (lt[3] == 1) {
            auto byte = [&](uint32_t b) { return (lt[b / 4] >> (8 * (b % 4))) & 255u; };
            for (uint32_t pass = 0; pass < 2; pass++) {
                const uint32_t b0 = pass ? MEL_MOM_SPLIT4_BYTE : MEL_MOM_SPLIT8_BYTE, nw = pass ? 4 : 8;
                if (byte(b0) != 0 || byte(b0 + nw) != h.n_groups) return -9;
                for (uint32_t w = 0; w < nw; w++)
                    if (byte(b0 + w) > byte(b0 + w + 1)) return -9;
            }
        } else if (h.n_groups <= 255) {
            return -9;
        }
    }
    return 0;
}

// The chirp-z plan (an odd factor of n_fft above 63; stft_bluestein_kernel): the kernel's phases (stft_core.h) run by one
// "thread" over the tables the plan uploads (bluestein_tables, host_math.cpp).  Every frame incl. the reflected boundary
// frames; out[f][k] = amplitude (not dB).  Returns M.
extern "C" __attribute__((visibility("default"))) int emu_stft_bluestein(const float *wav, uint32_t n_samples, uint32_t win, uint32_t hop,
                                                                          uint32_t n_fft, const float *window, uint32_t n_frames, float *out) {
    if (n_fft < 2 || n_fft % 2) return -1;
    StftGeom g{};
    g.hop = hop;
    g.win = win;
    g.n_fft = n_fft;
    g.pad_left = (n_fft - win) / 2;
    g.nc = n_fft / 2;
    g.n_freq = n_fft / 2 + 1;
    g.height = g.n_freq;
    uint32_t M = 1;
    while (M < 2 * g.nc - 1) M <<= 1;
    const BluesteinTables bt = bluestein_tables(n_fft, M);
    if (bt.chirp.size() != 2 * (size_t)g.nc || bt.bhat.size() != 2 * (size_t)M || bt.tws.size() != 2 * (size_t)(g.nc / 2 + 1)) return -2;
    const cf64 *chirp = reinterpret_cast<const cf64 *>(bt.chirp.data()), *bhat = reinterpret_cast<const cf64 *>(bt.bhat.data());
    const cf64 *twm = reinterpret_cast<const cf64 *>(bt.twm.data()), *tws = reinterpret_cast<const cf64 *>(bt.tws.data());
    std::vector<cf64> a(M), b(M);
    for (uint32_t f = 0; f < n_frames; f++) {
        const int64_t s0 = (int64_t)f * hop - (int64_t)(win / 2);
        bluestein_load(0, 1, g, M, wav, n_samples, s0, window, chirp, a.data());
        cf64 *in = a.data(), *o = b.data();
        for (int pass = 0; pass < 2; pass++) {
            for (uint32_t Ns = 1; Ns < M; Ns <<= 1) {
                bluestein_pass(0, 1, M, Ns, twm, in, o);
                std::swap(in, o);
            }
            if (pass == 0) bluestein_product(0, 1, M, bhat, in);
        }
        bluestein_unchirp(0, 1, g, M, chirp, in, o);
        bluestein_split(0, 1, g, tws, o, out + (size_t)f * g.n_freq);
    }
    return (int)M;
}

