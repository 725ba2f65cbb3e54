// mel_fuse.h — host-side tables for the mel filterbank fused into the wave STFT kernel's epilogue.
//
// Reference: `linspec.dot(&mel_fb)` (src-tauri/src/core/spectrogram.rs:207) with the Slaney filterbank of
// src-common/src/lib.rs:46-89.  The filters are triangles over consecutive centre frequencies, so every bin k has at
// most two non-zero weights: the falling edge of filter s-1 and the rising edge of filter s, where "segment" s is the
// run of bins between centre s and centre s+1.  The product is therefore 2 multiply-adds per bin, not a GEMM:
//     mel[m] = sum_{k in segment m} amp[k] * rise[k]  +  sum_{k in segment m+1} amp[k] * fall[k]
// For the 64-lane wave that owns a frame the segments are cut into PIECES of 4 consecutive bins (weights outside the
// segment are zero), one piece per lane and "slot"; a piece yields the partial sums (r, f), and mel m then adds the r
// of segment m's pieces and the f of segment m+1's.  Narrow segments (1-2 bins at low frequencies) and wide ones
// (tens of bins at the top) cost the same per lane this way, which a lane-per-filter loop does not manage.
//
// The table is built from the plan's actual filterbank values (not from the centre frequencies), and the structure it
// relies on is VERIFIED while building: if any bin feeds more than two filters, or two that are not neighbours, or the
// segments do not ascend, `ok` stays false and the plan uses the matrix-core path instead.
//
// Word layout (uint32, copied to LDS by the kernel), S = slots, G = groups of 64 mels:
//   k0  [64 S]        first bin of piece p = 64 slot + lane (always k0 + 4 <= n_freq)
//   w   [2 * 256 S]   ((slot * 4 + i) * 64 + lane) * 2 + {0: rise, 1: fall} weight of bin k0 + i      (float bits)
//   gat [64 G]        mel m = 64 g + lane: first piece | pieces of segment m << 16 | pieces of segment m+1 << 24
//   gmax[G + (G & 1)] max over the group's lanes of the two piece counts added (the gather loop's trip count)
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "stft_core.h"  // MEL_MOM_HDR0, MEL_MOM_BATCH

namespace th {

struct MelFuseHost {
    std::vector<uint32_t> words;
    uint32_t n_slots = 0, n_groups = 0, n_pieces = 0;
    bool ok = false;
};

constexpr uint32_t MEL_PIECE_BINS = 4;

// fb: [n_freq][n_mel] row-major (calc_mel_fb's layout).  max_pieces: capacity of the kernel's per-wave (r, f) buffer.
inline MelFuseHost build_mel_fuse(const float *fb, uint32_t n_freq, uint32_t n_mel, uint32_t max_pieces) {
    MelFuseHost out;
    if (n_freq < MEL_PIECE_BINS || n_mel == 0) return out;
    auto w = [&](uint32_t k, uint32_t m) { return fb[(size_t)k * n_mel + m]; };
    // segment of every bin (-1: the bin feeds no filter) and its two weights.  A bin under two filters m, m+1 is on
    // the falling edge of m and the rising edge of m+1: segment m+1.  A bin under ONE filter m may be filed as
    // "rising, segment m" or "falling, segment m+1" — its weight reaches mel m either way — whichever keeps the
    // segments in ascending order.
    std::vector<int32_t> seg(n_freq, -1);
    std::vector<float> rise(n_freq, 0.f), fall(n_freq, 0.f);
    int32_t last_seg = -1;
    for (uint32_t k = 0; k < n_freq; k++) {
        int32_t m1 = -1, m2 = -1;
        for (uint32_t m = 0; m < n_mel; m++) {
            const float v = w(k, m);
            if (v == 0.f) continue;
            if (!(v > 0.f)) return out;  // negative or NaN weight: not a triangle filterbank
            if (m1 < 0) m1 = (int32_t)m;
            else if (m2 < 0) m2 = (int32_t)m;
            else return out;  // three filters at one bin
        }
        if (m1 < 0) continue;
        int32_t s;
        if (m2 >= 0) {
            if (m2 != m1 + 1) return out;
            s = m2;
            fall[k] = w(k, (uint32_t)m1);
            rise[k] = w(k, (uint32_t)m2);
        } else if (last_seg <= m1) {
            s = m1;
            rise[k] = w(k, (uint32_t)m1);
        } else {
            s = m1 + 1;
            fall[k] = w(k, (uint32_t)m1);
        }
        if (s < last_seg) return out;  // segments must be runs of bins in ascending order
        last_seg = s;
        seg[k] = s;
    }
    // pieces, segment by segment; pb[s] = first piece of segment s (s = 0 .. n_mel; pb has n_mel + 2 entries)
    const uint32_t n_seg = n_mel + 1;
    std::vector<uint32_t> pb(n_seg + 1, 0), pk0;
    {
        std::vector<uint32_t> lo(n_seg, n_freq), hi(n_seg, 0);
        for (uint32_t k = 0; k < n_freq; k++)
            if (seg[k] >= 0) {
                lo[seg[k]] = std::min(lo[seg[k]], k);
                hi[seg[k]] = k + 1;
            }
        for (uint32_t s = 0; s < n_seg; s++) {
            pb[s] = (uint32_t)pk0.size();
            for (uint32_t k = lo[s]; k < hi[s]; k += MEL_PIECE_BINS) pk0.push_back(k);
        }
        pb[n_seg] = (uint32_t)pk0.size();
    }
    const uint32_t n_pieces = (uint32_t)pk0.size();
    if (n_pieces == 0 || n_pieces > max_pieces || n_pieces >= 65536) return out;
    const uint32_t S = (n_pieces + 63) / 64, G = (n_mel + 63) / 64;
    if (S > 8 || G > 8) return out;  // MEL_MAX_SLOTS / MEL_MAX_GROUPS of the lane functions (stft_wave.h)
    std::vector<uint32_t> &t = out.words;
    t.assign((size_t)64 * S + 512 * S + 64 * G + G + (G & 1), 0);
    uint32_t *k0 = t.data(), *wt = k0 + 64 * S, *gat = wt + 512 * S, *gmax = gat + 64 * G;
    // which segment a piece belongs to
    std::vector<uint32_t> pseg(n_pieces, 0);
    for (uint32_t s = 0; s < n_seg; s++)
        for (uint32_t p = pb[s]; p < pb[s + 1]; p++) pseg[p] = s;
    for (uint32_t p = 0; p < 64 * S; p++) {
        const uint32_t slot = p / 64, lane = p % 64;
        if (p >= n_pieces) {  // padding pieces: bins 0..3 against zero weights
            k0[p] = 0;
            continue;
        }
        const uint32_t first = pk0[p];
        const uint32_t base = std::min(first, n_freq - MEL_PIECE_BINS);  // keep the 4 reads inside the row
        k0[p] = base;
        for (uint32_t i = 0; i < MEL_PIECE_BINS; i++) {
            const uint32_t k = base + i;
            const bool mine = k >= first && k < first + MEL_PIECE_BINS && seg[k] == (int32_t)pseg[p];
            const float wr = mine ? rise[k] : 0.f, wf = mine ? fall[k] : 0.f;
            const size_t at = ((size_t)(slot * 4 + i) * 64 + lane) * 2;
            std::memcpy(&wt[at], &wr, 4);
            std::memcpy(&wt[at + 1], &wf, 4);
        }
    }
    for (uint32_t g = 0; g < G; g++) {
        uint32_t mx = 0;
        for (uint32_t lane = 0; lane < 64; lane++) {
            const uint32_t m = 64 * g + lane;
            if (m >= n_mel) continue;
            const uint32_t nr = pb[m + 1] - pb[m], nf = pb[m + 2] - pb[m + 1];
            if (nr > 255 || nf > 255) {
                t.clear();
                return out;
            }
            gat[m] = pb[m] | nr << 16 | nf << 24;
            mx = std::max(mx, nr + nf);
        }
        gmax[g] = mx;
    }
    out.n_slots = S;
    out.n_groups = G;
    out.n_pieces = n_pieces;
    out.ok = true;
    return out;
}

// ---------------------------------------------------------------------------------------------
// Banded-sum form of the same product (round 3): lane = mel.  Mel m = 64 g + lane adds amp[lo_m + t] * w_m[t] for
// t < n_g, n_g = the widest filter of its group of 64 mels rounded up to 4 (zero weights past a filter's own width; the
// caller keeps n_g finite floats readable behind the amplitude row).  The mels of a group are neighbours on the scale, so their
// widths are close (1-3 bins at the bottom, 14-25 in the last group of the 48 kHz default): the loop runs to the group's
// maximum and wastes little, and there is no second phase — no partial sums through LDS, no gather.  Works for any
// filterbank whose filters are narrow; no triangle structure is assumed.
// Words: [2 g] = offset of group g's block, [2 g + 1] = n_g (g < 8); block = 64 first bins, then n_g x 64 weights (float bits).
// ---------------------------------------------------------------------------------------------
struct MelBandHost {
    std::vector<uint32_t> words;
    uint32_t n_groups = 0, max_taps = 0;
    uint32_t taps_unshifted = 0;  // sum over the groups of their widest filter (rounded to 4), before the bank spreading below
    uint32_t reach = 0;  // one past the highest bin index any lane reads: max over the groups of (first bin + the group's taps)
    bool ok = false;
};
constexpr uint32_t MEL_BAND_MAX_GROUPS = 8, MEL_BAND_HDR = 2 * MEL_BAND_MAX_GROUPS, MEL_BAND_MAX_TAPS = 128;

// Bank spreading (round 4).  The amplitude reads of a tap are `ds_read_b32 amp[first_m + t]`, lane = mel: the LDS serves a
// wave's read as two halves of 32 lanes, bank = dword address mod 32, one more cycle for every further address on a busy bank
// (MI355X_MICROARCH.md, LDS).  The first bins of 32 neighbouring filters are spread quasi-randomly over the banks (3-4 on the
// busiest one for the wide filters: SQ_LDS_BANK_CONFLICT 27 % of SQ_LDS_IDX_ACTIVE at the 48 kHz default), and since every tap
// moves all lanes by the same amount the collisions are the same for every tap.  A filter narrower than its group's tap count
// has slack: it may start `s` bins early against `s` leading zero weights.  mel_band_spread picks the shifts of one half so
// that at most `cap` first bins fall on any bank (bipartite matching filter -> bank, smallest shifts tried first), within
// `taps`; returns false when no such assignment exists.
// step = 2 (paired layout, below): first bins must be even and a lane's 8-byte read covers the bank pair (first / 2) mod 32.
inline bool mel_band_spread(const uint32_t *lo, const uint32_t *width, uint32_t n, uint32_t taps, uint32_t cap, uint32_t *shift,
                            uint32_t step = 1) {
    // Kuhn's augmenting paths, filter -> (bank, one of its `cap` places); 32 x 32 cap, trivial sizes
    struct Rec {
        const uint32_t *lo, *width;
        uint32_t taps, cap, step;
        int owner[32][4];  // (bank, place) -> filter
        uint32_t pick[32];
        bool visit(uint32_t m, bool (&seen)[32]) {
            const uint32_t smax = std::min(lo[m], taps - width[m]);
            for (uint32_t s = (lo[m] % step); s <= smax; s += step) {
                const uint32_t b = ((lo[m] - s) / step) & 31u;
                if (seen[b]) continue;
                seen[b] = true;
                for (uint32_t c = 0; c < cap; c++)
                    if (owner[b][c] < 0 || visit((uint32_t)owner[b][c], seen)) {
                        owner[b][c] = (int)m;
                        pick[m] = s;
                        return true;
                    }
            }
            return false;
        }
    } rec{lo, width, taps, cap, step, {}, {}};
    if (n > 32 || cap < 1 || cap > 4) return false;
    for (auto &o : rec.owner) std::fill(o, o + 4, -1);
    for (uint32_t m = 0; m < n; m++) {
        bool seen[32] = {};
        if (width[m] > taps || !rec.visit(m, seen)) return false;
    }
    std::copy(rec.pick, rec.pick + n, shift);
    return true;
}
// LDS cycles of one tap's amplitude read for a half whose first bins are lo[0 .. n): the largest number of different
// addresses on one bank
inline uint32_t mel_band_half_cycles(const uint32_t *lo, uint32_t n) {
    uint32_t worst = n ? 1u : 0u;
    for (uint32_t b = 0; b < 32; b++) {
        std::vector<uint32_t> a;
        for (uint32_t m = 0; m < n; m++)
            if ((lo[m] & 31u) == b && std::find(a.begin(), a.end(), lo[m]) == a.end()) a.push_back(lo[m]);
        worst = std::max<uint32_t>(worst, (uint32_t)a.size());
    }
    return worst;
}

// Paired layout (round 4, `paired`): four taps are one 16-byte weight read (ds_read_b128, weights stored [tap / 4][lane][4])
// and two 8-byte amplitude reads (ds_read_b64 of amp[first + 2 j], first bins EVEN) instead of four + four dword reads: both
// wider reads move 256 B per LDS clock where ds_read_b32 / ds_read2_b32 move 128 (MI355X_MICROARCH.md, LDS), so a tap costs
// half the LDS cycles when the 32 lanes of a half sit on different bank pairs.  Making the first bins even and distinct
// mod 64 needs more slack than the dword spreading: a group's tap count may grow by 4 or 8.  `ok` stays false when no
// assignment with at most 4 addresses per bank exists (the caller then builds the plain layout).
inline MelBandHost build_mel_band(const float *fb, uint32_t n_freq, uint32_t n_mel, uint32_t max_words, bool spread = true,
                                  bool paired = false) {
    MelBandHost out;
    if (n_mel == 0 || n_freq == 0) return out;
    const uint32_t G = (n_mel + 63) / 64;
    if (G > MEL_BAND_MAX_GROUPS) return out;
    std::vector<uint32_t> lo(n_mel, 0), hi(n_mel, 0);
    for (uint32_t m = 0; m < n_mel; m++) {
        uint32_t a = n_freq, b = 0;
        for (uint32_t k = 0; k < n_freq; k++)
            if (fb[(size_t)k * n_mel + m] != 0.f) {
                a = std::min(a, k);
                b = k + 1;
            }
        lo[m] = b ? a : 0;
        hi[m] = b;
    }
    std::vector<uint32_t> &t = out.words;
    t.assign(MEL_BAND_HDR, 0);
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t m0 = 64 * g, m1 = std::min(n_mel, 64 * g + 64);
        uint32_t n = 0;
        for (uint32_t m = m0; m < m1; m++) n = std::max(n, hi[m] - lo[m]);
        n = std::max<uint32_t>(4, (n + 3) / 4 * 4);  // (mel_banded starts its sums with four taps)
        if (n > MEL_BAND_MAX_TAPS) {
            t.clear();
            return out;
        }
        out.taps_unshifted += n;
        // first bins: as they are, or moved down onto distinct banks where that costs fewer LDS cycles.  Four taps cost the
        // two halves' weight reads plus their amplitude reads — plain layout: 4 x (1 + 1) cycles + 4 x (the halves' busiest
        // banks, mel_band_half_cycles); paired: 4 + 2 x (the halves' busiest bank pairs).
        uint32_t shift[64] = {}, width[64] = {};
        for (uint32_t m = m0; m < m1; m++) width[m - m0] = hi[m] - lo[m];
        if (spread || paired) {
            const uint32_t h0 = std::min(32u, m1 - m0), h1 = m1 - m0 - h0;
            uint32_t best_cost = ~0u, best_n = n;
            if (!paired) best_cost = n * (2u + mel_band_half_cycles(&lo[m0], h0) + mel_band_half_cycles(&lo[m0] + h0, h1)) + n;
            const uint32_t step = paired ? 2u : 1u;
            auto try_shifts = [&](uint32_t nn, const uint32_t *sh) {  // the cost of an assignment by the bank rule; keeps the best
                uint32_t first[64] = {};
                for (uint32_t l = 0; l < h0 + h1; l++) first[l] = (lo[m0 + l] - sh[l]) / step;
                const uint32_t cycles = mel_band_half_cycles(first, h0) + mel_band_half_cycles(first + h0, h1);
                // (+ nn: a tap is also a multiply-add and its share of the loop, about one more cycle of the CU's time)
                const uint32_t cost = (paired ? nn / 4 * (4u + 2u * cycles) : nn * (2u + cycles)) + nn;
                if (cost < best_cost) {
                    best_cost = cost;
                    best_n = nn;
                    std::copy(sh, sh + 64, shift);
                }
            };
            for (uint32_t nn = n; nn <= std::min(n + 8u, MEL_BAND_MAX_TAPS); nn += 4) {
                uint32_t sh[64] = {};
                if (paired) {  // only made even: filters that start on the same bin (dense filterbanks) stay one broadcast address
                    bool fits = true;
                    for (uint32_t l = 0; l < h0 + h1; l++) {
                        sh[l] = lo[m0 + l] & 1u;
                        fits = fits && width[l] + sh[l] <= nn;
                    }
                    if (fits) try_shifts(nn, sh);
                }
                bool found = true;
                for (uint32_t h = 0; h < 2 && found; h++) {  // each half: the fewest filters per bank that can be had within nn taps
                    const uint32_t hn = h ? h1 : h0, hb = h ? h0 : 0;
                    if (!hn) continue;
                    uint32_t cap = 1;
                    while (cap <= 4 && !mel_band_spread(&lo[m0] + hb, width + hb, hn, nn, cap, sh + hb, step)) cap++;
                    found = cap <= 4;
                }
                if (found) try_shifts(nn, sh);
            }
            if (best_cost == ~0u) {  // (paired only)
                t.clear();
                return out;
            }
            n = best_n;
        }
        out.max_taps = std::max(out.max_taps, n);
        const size_t off = t.size();
        t[2 * g] = (uint32_t)off;
        t[2 * g + 1] = n;
        t.resize(off + 64 * (size_t)(1 + n), 0);
        for (uint32_t lane = 0; lane < 64; lane++) {
            // lanes past the last mel repeat the group's last filter (same addresses: served by the same broadcast; same sums: a
            // kernel may fold every lane's result into its min / max and mask only the store — round 5, mel_banded_pair's caller)
            const uint32_t src = 64 * g + lane < n_mel ? lane : n_mel - 1 - 64 * g;
            const uint32_t m = 64 * g + src;
            const uint32_t first = lo[m] - shift[src];
            t[off + lane] = first;
            out.reach = std::max(out.reach, first + n);
            for (uint32_t k = lo[m]; k < hi[m]; k++) {
                const uint32_t tap = k - first;
                const size_t at = paired ? off + 64 + ((size_t)(tap / 4) * 64 + lane) * 4 + tap % 4 : off + 64 * (size_t)(1 + tap) + lane;
                std::memcpy(&t[at], &fb[(size_t)k * n_mel + m], 4);
            }
        }
    }
    if (t.size() > max_words) {
        t.clear();
        return out;
    }
    out.n_groups = G;
    out.ok = true;
    return out;
}

// ---------------------------------------------------------------------------------------------
// Moment form of the same product (round 6): NO weight table.  The filters are triangles over the points p_0 < p_1 < ... <
// p_{n_mel + 1} (src-common/src/lib.rs:65-86: filter m rises over (p_m, p_{m+1}] and falls over (p_{m+1}, p_{m+2})), so on
// SEGMENT j = {bins k : p_j < f_k <= p_{j+1}} both weights are linear in the bin index,
//     rise_j[k] = u_k / d_j,   fall_{j-1}[k] = (1 - u_k) / d_{j-1},   u_k = (f_k - p_j) / (p_{j+1} - p_j) = alpha_j t + beta_j,
// t = k - (the lane's first bin), d_m = the filter's weight sum (lib.rs:84-86).  With the segment's moments
//     S0_j = sum_k amp_k,   S1_j = sum_k amp_k t:      R_j = alpha_j S1_j + beta_j S0_j,   F_j = S0_j - R_j,
//     mel[m] = (R_m + F_{m+1}) / d_m.
// Lane = segment (64 per group): per bin of a segment two additions (P += a; S1' += P, walking the taps downwards, S0 = P at the
// end; S1' = S1 + S0 and the host folds that into beta) where the banded sums spend a multiply-add per bin of a FILTER (two
// segments) plus the weight reads — and nothing to keep in LDS beside the amplitude row: the n_fft 4096 kernel's eight slabs
// leave no byte for a table (round 5 read a 36 KB one from L2 per frame: slower than two kernels).  Per group a lane loads four
// words (first bin, alpha, beta', 1 / d) from global memory (1 KiB per wave-load, L1 / L2 resident) and the wave one 64-bit
// lane mask per tap through the scalar cache: lanes whose segment does not cover tap t add zero.
//
// Two forms per group (word [2 + 2 g] bit 16):
//   M (moments, above) where the group's widest segment has three bins or more.  R and F are differences of terms of size
//     (alpha n + |beta'|) S0, so a filter whose bins all hug the far ends of its two segments (tiny d) would see that rounding
//     1 / d times enlarged: `max_amp` = max over the M groups of (alpha n + |beta'|) / d is checked (<= 4: the error stays at a
//     few ulp of the segment's amplitude sum over d, i.e. of a neighbouring mel value).
//   W (weights) where every segment of the group has at most two bins — the bottom of every default filterbank, where segments
//     are narrower than a bin and 1 / d reaches 10^4: the lane's one or two (u, 1 - u) pairs as the reference forms them in f32,
//     R = a0 u0 + a1 u1, F = a0 v0 + a1 v1: exact products, no masks, two or four operations per group.
//
// NOT bit-faithful to the table in the M groups: the reference's f32 weights carry the rounding of its f32 frequencies, the line
// through (p_j, p_{j+1}) does not.  `max_dev` is the largest |alpha t + beta - u_k(table)| over every bin of the M groups (1e-7
// .. 1e-5) and bounds the result's deviation from the table's product relative to the segment's amplitude sum over d.
//
// Words: [0] = G, [1] = sum of the groups' taps, [3] .. [8] the workgroup-per-frame kernels' ranges (mel_mom_splits); group g: [16 + 2 g] = taps n_g | form << 16 (M: n_g a multiple of
// MEL_MOM_UNROLL, zero masks on top; W: 1 or 2), [17 + 2 g] = word offset of its block (16-byte aligned); the header is padded to
// whole batches of MEL_MOM_BATCH groups, at least one batch, with (0, 0).  M block: 64 x {first
// bin, alpha, beta', 1 / d} in lane order, then n_g masks (2 words each, tap t at 2 t; mel_mom_mask_words(n_g) words), then 64
// WINDOW words, the masks once more per lane: first tap | taps << 16 (the workgroup-per-frame kernels' lockstep walk).  W block: 64 x {first bin, u0, v0, 1 / d},
// then 64 x {u1, v1}.  Lane l of group g = segment 64 g + l; mel m = 64 g + l takes F from the next lane (the next group's lane 0
// across the group border: the kernel walks the groups downwards and carries it).
// ---------------------------------------------------------------------------------------------
struct MelMomHost {
    std::vector<uint32_t> words;
    uint32_t n_groups = 0, taps = 0, max_taps = 0, reach = 0, w_groups = 0;
    double max_dev = 0.0, max_amp = 0.0;  // see above
    bool ok = false;
};
constexpr uint32_t MEL_MOM_UNROLL = 4, MEL_MOM_FORM_W = 1u << 16;
constexpr double MEL_MOM_MAX_DEV = 1.2e-5;  // (see build_mel_moments: banks beyond it keep a weight table)

// The workgroup-per-frame kernels share a frame's groups among their 4 or 8 waves in contiguous ranges; a wave walks its range
// [a, b) and the group b above it (for the F its lane 0 owes to mel 64 b - 1) in batches of MEL_MOM_BATCH groups counted from a,
// each batch in lockstep over the taps of its widest M group (mel_moments_range_lockstep, stft_wave.h).  The taps of a group
// grow with the mel index, so equal group counts leave the top waves with most of the work: the ranges are chosen here to
// minimise the largest wave's cost, with a batch priced at MEL_MOM_BATCH_COST + MEL_MOM_TRIP_COST per four taps of its walk
// (shader clocks measured with the cycle counter at n_fft 16384, 2785 mels: ~2000 per batch of W groups, ~900 per trip).
// Words [3] = 1, bytes 16.. = the 9 boundaries for 8 waves, bytes 28.. = the 5 boundaries for 4 waves.
constexpr uint32_t MEL_MOM_BATCH_COST = 2000, MEL_MOM_TRIP_COST = 900;  // (MEL_MOM_SPLIT8_BYTE, MEL_MOM_SPLIT4_BYTE: stft_core.h)
inline void mel_mom_splits(std::vector<uint32_t> &t, uint32_t G) {
    if (G == 0 || G > 255) return;  // ([3] stays 0: equal group counts)
    std::vector<uint32_t> n(G + 1, 0);  // taps of the M groups (W groups are not walked)
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t nw = t[MEL_MOM_HDR0 + 2 * g];
        n[g] = (nw & MEL_MOM_FORM_W) ? 0u : (nw & 0xffffu);
    }
    auto cost = [&](uint32_t a, uint32_t b) {
        if (a == b) return 0u;
        const uint32_t top = std::min(b + 1, G);
        uint32_t c = 0;
        for (uint32_t g0 = a; g0 < top; g0 += MEL_MOM_BATCH) {
            uint32_t run = 0;
            for (uint32_t g = g0; g < std::min(g0 + MEL_MOM_BATCH, top); g++) run = std::max(run, n[g]);
            c += MEL_MOM_BATCH_COST + MEL_MOM_TRIP_COST * ((run + 3) / 4);
        }
        return c;
    };
    auto put = [&](uint32_t waves, uint32_t byte0) {
        // best[w][g]: the smallest largest cost of the groups [0, g) in w ranges
        std::vector<std::vector<uint32_t>> best(waves + 1, std::vector<uint32_t>(G + 1, ~0u)), from(waves + 1, std::vector<uint32_t>(G + 1, 0));
        best[0][0] = 0;
        for (uint32_t w = 1; w <= waves; w++)
            for (uint32_t g = 0; g <= G; g++)
                for (uint32_t a = 0; a <= g; a++) {
                    if (best[w - 1][a] == ~0u) continue;
                    const uint32_t c = std::max(best[w - 1][a], cost(a, g));
                    if (c < best[w][g]) {
                        best[w][g] = c;
                        from[w][g] = a;
                    }
                }
        uint32_t g = G;
        for (uint32_t w = waves; w >= 1; w--) {
            const uint32_t byte = byte0 + w;
            t[byte / 4] |= g << (8 * (byte % 4));
            g = from[w][g];
        }
        // (boundary 0 = 0)
    };
    put(8, MEL_MOM_SPLIT8_BYTE);
    put(4, MEL_MOM_SPLIT4_BYTE);
    t[3] = 1;
}

// The table of the workgroup-per-frame kernels (n_fft 8192 / 16384), made from the one above: their waves all sit in the epilogue
// at once and a dependent load is a round trip to L2 that nothing hides (group header -> per-lane words -> the next batch's: five
// of them, 2.2 us of a 16384-point frame's 9), so a group's block sits at a FIXED address and holds everything per lane — plane 0:
// {first bin, alpha | u0, beta' | v0, 1 / d}, plane 1: {window | u1, v1, taps | form << 16, 0} — and a wave requests its first two
// batches in one go.  Words [0] .. [8] as above (G, taps, the waves' ranges); MEL_MOM_BATCH zero blocks behind the last group.
inline std::vector<uint32_t> build_mel_mom_lanes(const MelMomHost &h) {
    std::vector<uint32_t> t;
    if (!h.ok) return t;
    const uint32_t G = h.n_groups;
    t.assign(MEL_LANE_BLK0 + (size_t)MEL_LANE_STRIDE * (G + MEL_MOM_BATCH), 0);
    for (uint32_t i = 0; i < MEL_LANE_BLK0; i++) t[i] = h.words[i];
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t nw = h.words[MEL_MOM_HDR0 + 2 * g], off = h.words[MEL_MOM_HDR0 + 1 + 2 * g], n = nw & 0xffffu;
        const bool wform = (nw & MEL_MOM_FORM_W) != 0;
        uint32_t *const blk = &t[MEL_LANE_BLK0 + (size_t)MEL_LANE_STRIDE * g];
        for (uint32_t l = 0; l < 64; l++) {
            for (uint32_t i = 0; i < 4; i++) blk[4 * l + i] = h.words[off + 4 * l + i];
            if (wform) {
                blk[256 + 4 * l] = h.words[off + 256 + 2 * l];
                blk[256 + 4 * l + 1] = h.words[off + 256 + 2 * l + 1];
            } else {
                const uint32_t win = h.words[off + 256 + mel_mom_mask_words(n) + l];
                if ((win & 0xffffu) != 0) return {};  // (the lockstep walk takes a lane's window as [0, taps): a table built without spreading)
                blk[256 + 4 * l] = win;
            }
            blk[256 + 4 * l + 2] = nw;
        }
    }
    return t;
}

// fb: [n_freq][n_mel] (calc_mel_fb, normalised); lin[n_freq], mf[n_mel + 2]: the f32 frequency arrays it was built from
// (mel_fb_points); max_index: amplitude floats a lane may address (the wave's slab)
inline MelMomHost build_mel_moments(const float *fb, const float *lin, const float *mf, uint32_t n_freq, uint32_t n_mel,
                                    uint32_t max_index, bool spread = true) {
    MelMomHost out;
    if (n_mel == 0 || n_freq < 2) return out;
    const uint32_t n_seg = n_mel + 1, G = (n_seg + 63) / 64;
    // segments: runs of bins, ascending, each bin in at most one
    std::vector<uint32_t> lo(n_seg, 0), len(n_seg, 0);
    {
        uint32_t k = 0;
        while (k < n_freq && !(lin[k] > mf[0])) k++;
        for (uint32_t j = 0; j < n_seg; j++) {
            if (!(mf[j] < mf[j + 1])) return out;  // (coinciding points: the reference divides by zero there)
            lo[j] = k;
            while (k < n_freq && lin[k] <= mf[j + 1]) k++;
            len[j] = k - lo[j];
        }
        // every remaining bin must be weightless (f > fmax)
        for (; k < n_freq; k++)
            for (uint32_t m = 0; m < n_mel; m++)
                if (fb[(size_t)k * n_mel + m] != 0.f) return out;
    }
    auto u_f32 = [&](uint32_t j, uint32_t k) -> float {  // the reference's f32 rise weight of filter j at bin k of segment j
        return lin[k] == mf[j + 1] ? 1.f : (lin[k] - mf[j]) / (mf[j + 1] - mf[j]);
    };
    auto v_f32 = [&](uint32_t j, uint32_t k) -> float {  // ... and the fall weight of filter j - 1 there
        return lin[k] == mf[j + 1] ? 0.f : (mf[j + 1] - lin[k]) / (mf[j + 1] - mf[j]);
    };
    // d_m as the reference sums it (f32, ascending bins, lib.rs:84-86), cross-checked against the table value by value
    std::vector<float> dsum(n_mel, 0.f);
    for (uint32_t m = 0; m < n_mel; m++) {
        float s = 0.f;
        for (uint32_t k = lo[m]; k < lo[m] + len[m]; k++) s += u_f32(m, k);
        for (uint32_t k = lo[m + 1]; k < lo[m + 1] + len[m + 1]; k++) s += v_f32(m + 1, k);
        const float d = std::max(s, 1.1920929e-7f);
        dsum[m] = d;
        for (uint32_t k = lo[m]; k < lo[m] + len[m]; k++)
            if (fb[(size_t)k * n_mel + m] != u_f32(m, k) / d) return out;
        for (uint32_t k = lo[m + 1]; k < lo[m + 1] + len[m + 1]; k++)
            if (fb[(size_t)k * n_mel + m] != v_f32(m + 1, k) / d) return out;
    }
    const double step = (double)lin[1] - (double)lin[0];
    std::vector<uint32_t> &t = out.words;
    const uint32_t hdr = MEL_MOM_HDR0 + 2 * ((G + 2 * MEL_MOM_BATCH - 1) / MEL_MOM_BATCH * MEL_MOM_BATCH);  // (a batch of padding groups behind the last one: taps 0, offset 0 — a kernel's batch may start at any group)
    t.assign(hdr, 0);
    t[0] = G;
    for (uint32_t g = 0; g < G; g++) {
        const uint32_t s0 = 64 * g, s1 = std::min(n_seg, 64 * g + 64), ns = s1 - s0;
        uint32_t n = 1;
        for (uint32_t j = s0; j < s1; j++) n = std::max(n, len[j]);
        const size_t off = t.size();
        t[MEL_MOM_HDR0 + 1 + 2 * g] = (uint32_t)off;
        if (n <= 2) {  // W form
            t[MEL_MOM_HDR0 + 2 * g] = n | MEL_MOM_FORM_W;
            out.taps += n;
            out.max_taps = std::max(out.max_taps, n);
            out.w_groups++;
            t.resize(off + 256 + 128, 0);
            for (uint32_t l = 0; l < 64; l++) {
                const uint32_t j = s0 + l;
                float w[4] = {0.f, 0.f, 0.f, 0.f};  // u0, v0, u1, v1
                uint32_t first = 0;
                if (j < n_seg && len[j] > 0) {
                    first = lo[j];
                    for (uint32_t i = 0; i < len[j]; i++) {
                        if (j < n_mel) w[2 * i] = u_f32(j, lo[j] + i);  // (the last segment has no rising filter)
                        if (j >= 1) w[2 * i + 1] = v_f32(j, lo[j] + i);  // (the first none that falls)
                    }
                    out.reach = std::max(out.reach, first + n);
                }
                if (first + n > max_index) return MelMomHost{};
                const float inv = j < n_mel ? (float)(1.0 / (double)dsum[j]) : 0.f;
                std::memcpy(&t[off + 4 * l], &first, 4);
                std::memcpy(&t[off + 4 * l + 1], &w[0], 8);
                std::memcpy(&t[off + 4 * l + 3], &inv, 4);
                std::memcpy(&t[off + 256 + 2 * l], &w[2], 8);
            }
            continue;
        }
        // M form.  First bins moved down onto different LDS banks where the group's tap count leaves room (mel_band_spread: at
        // most `cap` first bins of a half wave on one bank); the masks cut the window out, so a shift costs nothing but beta
        uint32_t shift[64] = {};
        if (spread && ns > 1) {
            uint32_t width[64] = {}, first[64] = {};
            for (uint32_t l = 0; l < ns; l++) {
                width[l] = std::max(1u, len[s0 + l]);
                first[l] = lo[s0 + l];
            }
            const uint32_t h0 = std::min(32u, ns), h1 = ns - h0;
            // (cost of a tap: three vector operations + the LDS cycles of its amplitude read by the bank rule)
            uint32_t best = n * (3u + mel_band_half_cycles(first, h0) + (h1 ? mel_band_half_cycles(first + 32, h1) : 0u)), best_n = n;
            for (uint32_t nn = n; nn <= n + 2; nn++) {
                uint32_t sh[64] = {};
                bool found = true;
                uint32_t cycles = 0;
                for (uint32_t h = 0; h < 2 && found; h++) {
                    const uint32_t hb = h ? 32u : 0u, hn = h ? h1 : h0;
                    if (!hn) continue;
                    uint32_t cap = 1;
                    while (cap <= 4 && !mel_band_spread(first + hb, width + hb, hn, nn, cap, sh + hb, 1)) cap++;
                    found = cap <= 4;
                    cycles += cap;
                }
                if (found && nn * (3u + cycles) < best) {
                    best = nn * (3u + cycles);
                    best_n = nn;
                    std::copy(sh, sh + 64, shift);
                }
            }
            // a shift enlarges beta' and with it the rounding a small d multiplies (max_amp): not in a group that holds such a filter
            double amp_shifted = 0.0;
            for (uint32_t l = 0; l < ns; l++) {
                const uint32_t j = s0 + l;
                if (!len[j]) continue;
                const double dp = (double)mf[j + 1] - (double)mf[j], alpha = step / dp;
                const double beta = ((double)lin[0] + step * (double)(lo[j] - shift[l]) - (double)mf[j]) / dp - alpha;
                const double mag = alpha * (double)(shift[l] + len[j]) + std::fabs(beta);
                if (j < n_mel) amp_shifted = std::max(amp_shifted, mag / (double)dsum[j]);
                if (j >= 1) amp_shifted = std::max(amp_shifted, (1.0 + mag) / (double)dsum[j - 1]);
            }
            if (amp_shifted > 4.0) {
                std::fill(shift, shift + 64, 0u);
                best_n = 1;
                for (uint32_t j = s0; j < s1; j++) best_n = std::max(best_n, len[j]);
            }
            n = best_n;
        }
        const uint32_t n_pad = (n + MEL_MOM_UNROLL - 1) / MEL_MOM_UNROLL * MEL_MOM_UNROLL;
        t[MEL_MOM_HDR0 + 2 * g] = n_pad;
        out.taps += n_pad;
        out.max_taps = std::max(out.max_taps, n_pad);
        const size_t mw = mel_mom_mask_words(n_pad);
        t.resize(off + 256 + mw + 64, 0);  // (the kernel's batch fetch reads 128 words behind every block's per-lane words)
        for (uint32_t l = 0; l < 64; l++) {
            const uint32_t j = s0 + l;
            float prm[4] = {0.f, 0.f, 0.f, 0.f};
            uint32_t first = 0;
            if (j < n_seg && len[j] > 0) {
                first = lo[j] - shift[l];
                if (first + n_pad > max_index) return MelMomHost{};
                const double dp = (double)mf[j + 1] - (double)mf[j];
                const double alpha = step / dp;
                const double beta = ((double)lin[0] + step * (double)first - (double)mf[j]) / dp;  // u at tap 0
                prm[1] = (float)alpha;
                prm[2] = (float)(beta - alpha);  // the kernel's S1' counts tap t (t + 1) times: S1' = S1 + S0
                for (uint32_t k = lo[j]; k < lo[j] + len[j]; k++) {
                    const uint32_t tap = k - first;
                    const uint64_t bit = 1ull << l;
                    t[off + 256 + 2 * tap] |= (uint32_t)bit;
                    t[off + 256 + 2 * tap + 1] |= (uint32_t)(bit >> 32);
                    // deviation of the line (as the kernel's f32 constants give it) from the table's weights
                    const double u_line = (double)prm[1] * (double)(tap + 1) + (double)prm[2];
                    if (j < n_mel) out.max_dev = std::max(out.max_dev, std::fabs(u_line - (double)u_f32(j, k)));
                    if (j >= 1) out.max_dev = std::max(out.max_dev, std::fabs((1.0 - u_line) - (double)v_f32(j, k)));
                }
                const double mag = (double)prm[1] * (double)(shift[l] + len[j]) + std::fabs((double)prm[2]);
                if (j < n_mel) out.max_amp = std::max(out.max_amp, mag / (double)dsum[j]);
                if (j >= 1) out.max_amp = std::max(out.max_amp, (1.0 + mag) / (double)dsum[j - 1]);
                out.reach = std::max(out.reach, first + n_pad);
            }
            if (j < n_mel) prm[3] = (float)(1.0 / (double)dsum[j]);
            std::memcpy(&t[off + 4 * l], &first, 4);
            std::memcpy(&t[off + 4 * l + 1], &prm[1], 12);
            if (j < n_seg && len[j] > 0) {
                if (shift[l] > 0xffffu || len[j] > 0xffffu) return MelMomHost{};
                t[off + 256 + mw + l] = shift[l] | len[j] << 16;  // the lane's window: what the masks say, per lane
            }
        }
    }
    t.resize(t.size() + 384, 0);  // (padding groups fetch "block" 0, the header: 256 + 128 words must exist behind word 0 in every table)
    t[1] = out.taps;
    out.n_groups = G;
    mel_mom_splits(t, G);
    // (a filterbank whose segments are not lines in the bin index, or whose wide groups hold a filter with next to no weight: not this form)
    // The lines may leave the table's weights by MEL_MOM_MAX_DEV at most.  The deviation is the reference's own rounding of its f32
    // bin frequencies relative to a segment's width: 1e-7 where the bin spacing is an f32 number with room for the bin index (48 /
    // 96 / 192 kHz), 0.5 - 1.1e-5 for the 44.1 / 88.2 kHz defaults at n_fft 4096 and 88.2 kHz at 8192 — and 2 - 4e-5 for the finer
    // banks of that family (44.1 kHz at n_fft 8192 / 16384, 1000+ mels at 4096), where a filter output at 1 % of a frame's maximum
    // was seen 2e-3 dB off the table's (the parity tests allow 1e-3): those keep the table, i.e. the two kernels.
    out.ok = out.max_dev <= MEL_MOM_MAX_DEV && out.max_amp <= 8.0;
    if (!out.ok) t.clear();
    return out;
}

}  // namespace th
