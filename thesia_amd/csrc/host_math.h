// host_math.h — plan-preparation arithmetic done once on the host, exactly as the reference's
// SpectrogramAnalyzer::prepare does (spectrogram.rs:116-154): window and mel filterbank tables,
// framing parameters, tile geometry.  These are tables/shape decisions, not the hot path.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace th {

void calc_framing_params(double win_ms, uint32_t t_overlap, uint32_t f_overlap, uint32_t sr, size_t *hop,
                         size_t *win, size_t *n_fft);
size_t stft_n_frames(size_t n, size_t win, size_t hop);
std::vector<float> normalized_hann(size_t win, size_t n_fft);
float mel_from_hz(float hz);
float mel_to_hz(float mel);
std::vector<float> calc_mel_fb(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, bool do_norm);
void mel_fb_points(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, std::vector<float> &lin, std::vector<float> &mf);
size_t mel_default_n_mel(uint32_t sr, size_t n_fft);
void hz_range_to_idx(int freq_scale, float hz0, float hz1, uint32_t sr, size_t n, size_t *i0, size_t *i1);
void shard_assign(const uint64_t *weights, size_t n, uint32_t world, uint32_t *owner);
void global_db_range(const float *mins, const float *maxs, size_t n, float dB_range, float *mn, float *mx);

// Tables of a Bluestein plan (stft_bluestein_kernel, kernels_stft.hip): interleaved (re, im) doubles.  nc = n_fft / 2 points,
// M = 2^m >= 2 nc - 1.  chirp[n] = e^{-i pi n^2 / nc} (n^2 reduced mod 2 nc before the angle is formed), bhat = FFT_M of
// b[n] = b[M - n] = conj chirp[n], twm[k] = e^{-2 pi i k / M} (k < M / 2), tws[k] = e^{-2 pi i k / n_fft} (k <= nc / 2).
struct BluesteinTables {
    std::vector<double> chirp, bhat, twm, tws;
};
BluesteinTables bluestein_tables(size_t n_fft, size_t M);

struct TileGeom {
    size_t width, height, origin_x, origin_y, lod_w, lod_h;
};
TileGeom spectrogram_tile_geometry(size_t W, size_t Hh, uint32_t lx, uint32_t ly, uint32_t tx, uint32_t ty);
void waveform_tile_geometry(size_t n, uint32_t level, uint32_t tile, size_t *start, size_t *bins, size_t *spb);

// Lanczos3 tap table of one axis of resize_spectrogram_tile (render_tiles.rs:354-393; PARITY UNPINNED against
// fast_image_resize 6.0.0, whose source is not vendored: the textbook filter, exactly as oracle/thesia_oracle.c builds
// it).  n_out outputs over the source interval [origin, origin + extent); output o: taps for the source indices
// start[o] .. start[o] + count[o] - 1 (clamped to [lo, hi)), weights w[o * max_taps + t], their sum wsum[o].
struct LodAxisHost {
    std::vector<int32_t> start, count;
    std::vector<double> wsum, w;
    uint32_t max_taps = 0;
    size_t blob_bytes(size_t n_out) const { return n_out * 16 + n_out * (size_t)max_taps * 8; }
    void pack(unsigned char *p, size_t n_out) const;  // [start i32][count i32][wsum f64][w f64], as LodAxis reads it
};
void build_lod_axis(double origin, double extent, size_t n_out, long lo, long hi, LodAxisHost &ax);

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline unsigned ilog2(size_t n) {
    unsigned l = 0;
    while ((size_t(1) << (l + 1)) <= n) l++;
    return l;
}

}  // namespace th
