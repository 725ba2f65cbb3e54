// host_math.cpp — see host_math.h.  Each function cites the reference code it mirrors
// (paths relative to the reference checkout).
#include "host_math.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace th {

// SpecSetting::calc_framing_params — src-tauri/src/core/spectrogram.rs:56-98
void calc_framing_params(double win_ms, uint32_t t_overlap, uint32_t f_overlap, uint32_t sr, size_t *hop,
                         size_t *win, size_t *n_fft) {
    const double win_float = win_ms * double(sr) / 1000.;                  // calc_win_length_float
    const double h = std::round(win_float / double(t_overlap));           // f64::round: half away from 0
    const size_t hop_length = h > 0. ? size_t(h) : 0;                     // `as usize` saturates
    const size_t win_length = hop_length * size_t(t_overlap);
    size_t p = 1;
    while (p < win_length) p <<= 1;                                       // next_power_of_two
    *hop = hop_length;
    *win = win_length;
    *n_fft = p * size_t(f_overlap);
}

// Frame count of perform_stft — stft.rs:50-97.  The three-segment construction (front / mid /
// back) yields exactly the frames of a win/2 reflect-padded signal: T = (N + 2*(win/2) - win)/hop + 1
// (tests/test_oracle_numpy.py checks this against the literal restatement).
size_t stft_n_frames(size_t n, size_t win, size_t hop) {
    if (!n || !win || !hop) return 0;
    const size_t padded = n + 2 * (win / 2);
    if (padded < win) return 0;
    return (padded - win) / hop + 1;
}

// calc_normalized_win(Hann, size, n_fft) — windows.rs:12-38,68-83, evaluated in f32 in the
// reference's operation order: x = PI*i/size; (0.5 - 0.5*cos(2x)) + (0*cos(4x) - 0*cos(6x)); / n_fft
std::vector<float> normalized_hann(size_t win, size_t n_fft) {
    std::vector<float> w(win);
    const float pi = 3.14159265358979323846f;
    const float norm = float(n_fft);
    for (size_t i = 0; i < win; i++) {
        const float x = pi * float(i) / float(win);  // size2 - 1 == size for the periodic window
        const float b_ = 0.5f * std::cos(2.0f * x);
        const float c_ = 0.0f * std::cos(4.0f * x);
        const float d_ = 0.0f * std::cos(6.0f * x);
        w[i] = ((0.5f - b_) + (c_ - d_)) / norm;
    }
    return w;
}

// Slaney mel scale — src-common/src/lib.rs:11-43 (f32 instantiation)
static const float kMinLogHz = 1000.f;
static const float kMinLogMel = 15.f;
static const float kLogStep = float(0.06875177742094912);
static const float kLinearScale = float(200. / 3.);

float mel_to_hz(float mel) {
    if (mel < kMinLogMel) return kLinearScale * mel;
    return kMinLogHz * std::exp(kLogStep * (mel - kMinLogMel));
}
float mel_from_hz(float hz) {
    if (hz < kMinLogHz) return hz / kLinearScale;
    return kMinLogMel + std::log(hz / kMinLogHz) / kLogStep;
}

static void linspace(float a, float b, size_t n, float *out) {  // ndarray Array::linspace
    const float step = n > 1 ? (b - a) / float(n - 1) : 0.f;
    for (size_t i = 0; i < n; i++) out[i] = a + step * float(i);
}

// the two f32 frequency arrays of calc_mel_fb (src-common/src/lib.rs:61-67): the bins' frequencies and the n_mel + 2 triangle
// points; fmax < 0 = Nyquist
void mel_fb_points(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, std::vector<float> &lin, std::vector<float> &mf) {
    const float f_nyquist = float(double(sr) / 2.);
    if (fmax < 0.f) fmax = f_nyquist;
    const size_t n_freq = n_fft / 2 + 1;
    lin.assign(n_freq, 0.f);
    mf.assign(n_mel + 2, 0.f);
    linspace(0.f, f_nyquist, n_freq, lin.data());
    linspace(mel_from_hz(fmin), mel_from_hz(fmax), n_mel + 2, mf.data());
    for (auto &m : mf) m = mel_to_hz(m);
}

// calc_mel_fb::<f32> — src-common/src/lib.rs:46-89; returns (n_fft/2+1) x n_mel, C order
std::vector<float> calc_mel_fb(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, bool do_norm) {
    const size_t n_freq = n_fft / 2 + 1;
    std::vector<float> lin, mf, w(n_freq), fb(n_freq * n_mel, 0.f);
    mel_fb_points(sr, n_fft, n_mel, fmin, fmax, lin, mf);
    for (size_t m = 0; m < n_mel; m++) {
        std::fill(w.begin(), w.end(), 0.f);
        for (size_t i = 0; i < n_freq; i++) {
            const float f = lin[i];
            if (f <= mf[m]) continue;
            if (mf[m] < f && f < mf[m + 1]) w[i] = (f - mf[m]) / (mf[m + 1] - mf[m]);
            else if (f == mf[m + 1]) w[i] = 1.f;
            else if (mf[m + 1] < f && f < mf[m + 2]) w[i] = (mf[m + 2] - f) / (mf[m + 2] - mf[m + 1]);
            else break;
        }
        if (do_norm) {
            float s = 0.f;
            for (size_t i = 0; i < n_freq; i++) s += w[i];
            const float d = std::max(s, std::numeric_limits<float>::epsilon());
            for (size_t i = 0; i < n_freq; i++) w[i] /= d;
        }
        for (size_t i = 0; i < n_freq; i++) fb[i * n_mel + m] = w[i];  // weights.t()
    }
    return fb;
}

// calc_mel_fb_default's n_mel search — src-common/src/lib.rs:91-103
size_t mel_default_n_mel(uint32_t sr, size_t n_fft) {
    const float r = mel_from_hz(float(sr) / 2.f) / mel_from_hz(float(sr) / float(n_fft));
    const float v = std::fma(r, 2.f, -1.f);
    size_t n_mel = (v > 0.f) ? size_t(v) : 0;
    const size_t n_freq = n_fft / 2 + 1;
    n_mel = std::min(n_mel, n_freq);
    while (n_mel > 0) {
        const std::vector<float> fb = calc_mel_fb(sr, n_fft, n_mel, 0.f, -1.f, true);
        bool ok = true;
        for (size_t m = 0; m < n_mel && ok; m++) {
            float s = 0.f;
            for (size_t i = 0; i < n_freq; i++) s += fb[i * n_mel + m];
            ok = s > 0.f;
        }
        if (ok) break;
        n_mel--;
    }
    return n_mel;
}

// FreqScale::hz_range_to_idx — src-common/src/lib.rs:134-159
void hz_range_to_idx(int freq_scale, float hz0, float hz1, uint32_t sr, size_t n, size_t *i0, size_t *i1) {
    if (hz0 >= hz1) {
        *i0 = *i1 = 0;
        return;
    }
    const float half_sr = float(sr) / 2.f;
    const float r0 = freq_scale ? mel_from_hz(hz0) / mel_from_hz(half_sr) : hz0 / half_sr;
    const float r1 = freq_scale ? mel_from_hz(hz1) / mel_from_hz(half_sr) : hz1 / half_sr;
    float lo = std::floor(r0 * float(n));
    if (!(lo > 0.f)) lo = 0.f;
    const float hi = std::ceil(r1 * float(n));
    *i0 = size_t(lo);
    *i1 = hi > 0.f ? size_t(hi) : 0;
}

static inline float rs_min(float a, float b) { return std::isnan(a) ? b : (std::isnan(b) ? a : (a < b ? a : b)); }
static inline float rs_max(float a, float b) { return std::isnan(a) ? b : (std::isnan(b) ? a : (a > b ? a : b)); }

// TrackManager::update_spec_imgs range clamp — core/mod.rs:169-180
void global_db_range(const float *mins, const float *maxs, size_t n, float dB_range, float *mn, float *mx) {
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = 0; i < n; i++) {
        lo = rs_min(lo, mins[i]);
        hi = rs_max(hi, maxs[i]);
    }
    hi = rs_min(hi, 0.f);
    lo = rs_max(lo, hi - dB_range);
    *mn = lo;
    *mx = hi;
}

// Longest-processing-time-first assignment: units sorted by weight (descending, index ascending
// on ties) go to the currently least-loaded rank (lowest rank on ties).  Equal weights reduce to
// round-robin, e.g. 1024 equal tracks on 8 GPUs = 128 per GPU.
void shard_assign(const uint64_t *weights, size_t n, uint32_t world, uint32_t *owner) {
    std::vector<size_t> order(n);
    for (size_t i = 0; i < n; i++) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return weights[a] > weights[b]; });
    std::vector<uint64_t> load(world, 0);
    for (size_t i : order) {
        uint32_t best = 0;
        for (uint32_t r = 1; r < world; r++)
            if (load[r] < load[best]) best = r;
        owner[i] = best;
        load[best] += weights[i];
    }
}

static inline size_t sat_mul(size_t a, size_t b) {
    size_t r;
    return __builtin_mul_overflow(a, b, &r) ? SIZE_MAX : r;
}
static inline size_t sat_add(size_t a, size_t b) {
    size_t r;
    return __builtin_add_overflow(a, b, &r) ? SIZE_MAX : r;
}
static inline size_t shl_or_max(uint32_t level) { return level < 64 ? (size_t(1) << level) : SIZE_MAX; }
static inline size_t div_ceil(size_t a, size_t b) { return a / b + (a % b != 0); }

// encode_spectrogram_tile geometry — render_tiles.rs:290-313
TileGeom spectrogram_tile_geometry(size_t W, size_t Hh, uint32_t lx, uint32_t ly, uint32_t tx, uint32_t ty) {
    const size_t T = 512, G = 4;
    TileGeom g{};
    g.lod_w = div_ceil(W, shl_or_max(lx));
    g.lod_h = div_ceil(Hh, shl_or_max(ly));
    const size_t sx = sat_mul(tx, T), sy = sat_mul(ty, T);
    const size_t cw = std::min(g.lod_w > sx ? g.lod_w - sx : 0, T);
    const size_t ch = std::min(g.lod_h > sy ? g.lod_h - sy : 0, T);
    g.origin_x = sx > G ? sx - G : 0;
    g.origin_y = sy > G ? sy - G : 0;
    if (cw && ch) {
        const size_t rx = std::min(g.lod_w, sx + cw + G), ry = std::min(g.lod_h, sy + ch + G);
        g.width = rx > g.origin_x ? rx - g.origin_x : 0;
        g.height = ry > g.origin_y ? ry - g.origin_y : 0;
    }
    return g;
}

// encode_waveform_tile geometry — render_tiles.rs:233-241
void waveform_tile_geometry(size_t n, uint32_t level, uint32_t tile, size_t *start, size_t *bins, size_t *spb) {
    const size_t s = shl_or_max(level);
    const size_t tile_samples = sat_mul(1024, s);
    const size_t st = sat_mul(size_t(tile), tile_samples);
    const size_t en = std::min(n, sat_add(st, tile_samples));
    *start = st;
    *spb = s;
    *bins = st >= en ? 0 : div_ceil(en - st, s);
}

// Lanczos3 taps in the arithmetic of the convolution fast_image_resize documents itself as following (Pillow's
// ImagingResample, precompute_coeffs): lanczos(t) = sinc(t) sinc(t / 3) on -3 <= t < 3, sinc(t) = sin(pi t) / (pi t).
static double pil_sinc(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return std::sin(x) / x;
}
static double pil_lanczos(double x) {
    if (-3.0 <= x && x < 3.0) return pil_sinc(x) * pil_sinc(x / 3);
    return 0.0;
}

// Window [ (int)(center - support + 0.5), (int)(center + support + 0.5) ) clipped at [lo, hi) = the IMAGE, tap k =
// lanczos(((k + xmin) - center + 0.5) * (1 / filterscale)), taps divided by their sum HERE (one division per tap, not per
// pixel): the kernels then add pixel * tap in ascending order from 0.0 and round half up, which makes a level bit-identical
// to Pillow 12.2's 16-bit Lanczos resize (tests/golden/lod_pillow_cases.npz) wherever that does not overflow 65535.
// wsum stays in the table's layout as 1.0 (x / 1.0 == x exactly).
void build_lod_axis(double origin, double extent, size_t n_out, long lo, long hi, LodAxisHost &ax) {
    const double sc = extent / (double)n_out, f = sc < 1.0 ? 1.0 : sc, sup = 3.0 * f, ss = 1.0 / f;
    ax.start.resize(n_out);
    ax.count.resize(n_out);
    ax.wsum.assign(n_out, 1.0);
    ax.max_taps = 0;
    for (size_t o = 0; o < n_out; o++) {
        const double center = origin + ((double)o + 0.5) * sc;
        long i0 = (long)(center - sup + 0.5), i1 = (long)(center + sup + 0.5);  // (truncation, as the C casts of Pillow)
        if (i0 < lo) i0 = lo;
        if (i1 > hi) i1 = hi;
        if (i1 < i0) i1 = i0;
        ax.start[o] = (int32_t)i0;
        ax.count[o] = (int32_t)(i1 - i0);
        ax.max_taps = std::max<uint32_t>(ax.max_taps, (uint32_t)(i1 - i0));
    }
    if (ax.max_taps == 0) ax.max_taps = 1;
    ax.w.assign(n_out * (size_t)ax.max_taps, 0.0);
    for (size_t o = 0; o < n_out; o++) {
        const double center = origin + ((double)o + 0.5) * sc;
        double *const k = &ax.w[o * (size_t)ax.max_taps];
        double ww = 0.0;
        for (int32_t t = 0; t < ax.count[o]; t++) {
            const double wv = pil_lanczos(((double)(t + ax.start[o]) - center + 0.5) * ss);
            k[t] = wv;
            ww += wv;
        }
        if (ww != 0.0)
            for (int32_t t = 0; t < ax.count[o]; t++) k[t] /= ww;
    }
}

void LodAxisHost::pack(unsigned char *p, size_t n_out) const {
    std::memcpy(p, start.data(), n_out * 4);
    std::memcpy(p + n_out * 4, count.data(), n_out * 4);
    std::memcpy(p + n_out * 8, wsum.data(), n_out * 8);
    std::memcpy(p + n_out * 16, w.data(), w.size() * 8);
}


BluesteinTables bluestein_tables(size_t n_fft, size_t M) {
    BluesteinTables t;
    const size_t nc = n_fft / 2, h = M / 2;
    t.chirp.resize(2 * nc);
    for (size_t n = 0; n < nc; n++) {
        const uint64_t r = (uint64_t)n * n % (2 * (uint64_t)nc);  // e^{-i pi n^2 / nc} has period 2 nc in n^2
        const double a = -M_PI * (double)r / (double)nc;
        t.chirp[2 * n] = std::cos(a);
        t.chirp[2 * n + 1] = std::sin(a);
    }
    t.twm.resize(2 * std::max<size_t>(h, 1));
    for (size_t k = 0; k < std::max<size_t>(h, 1); k++) {
        const double a = -2.0 * M_PI * (double)k / (double)M;
        t.twm[2 * k] = std::cos(a);
        t.twm[2 * k + 1] = std::sin(a);
    }
    t.tws.resize(2 * (nc / 2 + 1));
    for (size_t k = 0; k <= nc / 2; k++) {
        const double a = -2.0 * M_PI * (double)k / (double)n_fft;
        t.tws[2 * k] = std::cos(a);
        t.tws[2 * k + 1] = std::sin(a);
    }
    // bhat = FFT_M(b): radix-2 Stockham, the kernel's own pass structure
    std::vector<double> x(2 * M, 0.0), y(2 * M, 0.0);
    for (size_t n = 0; n < nc; n++) {
        x[2 * n] = t.chirp[2 * n];
        x[2 * n + 1] = -t.chirp[2 * n + 1];
        if (n) {
            x[2 * (M - n)] = t.chirp[2 * n];
            x[2 * (M - n) + 1] = -t.chirp[2 * n + 1];
        }
    }
    for (size_t Ns = 1; Ns < M; Ns <<= 1) {
        const size_t step = h / Ns;
        for (size_t j = 0; j < h; j++) {
            const size_t k = j & (Ns - 1), j0 = (j - k) * 2 + k;
            const double wr = t.twm[2 * k * step], wi = t.twm[2 * k * step + 1];
            const double ar = x[2 * j], ai = x[2 * j + 1], br0 = x[2 * (j + h)], bi0 = x[2 * (j + h) + 1];
            const double br = br0 * wr - bi0 * wi, bi = br0 * wi + bi0 * wr;
            y[2 * j0] = ar + br;
            y[2 * j0 + 1] = ai + bi;
            y[2 * (j0 + Ns)] = ar - br;
            y[2 * (j0 + Ns) + 1] = ai - bi;
        }
        x.swap(y);
    }
    t.bhat = std::move(x);
    return t;
}

}  // namespace th
