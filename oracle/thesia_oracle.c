/*
 * thesia_oracle.c — CPU restatement of thesia's spectrogram / waveform hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker / reported CPU baseline.  The
 * product (thesia_amd/) never links, imports or calls it.
 *
 * The reference is Rust and cannot be built here (no cargo/rustc, crates not
 * vendored).  Every function below restates one reference function in plain C
 * and cites the reference file:line it follows (paths relative to the
 * reference checkout).  Parity pinning: tests/test_oracle_golden.py checks this
 * file against every known-answer test the reference holds for the path
 * (SURVEY.md §4 / §8c).  Arithmetic that lives in un-vendored crates:
 *   - realfft 3.5.0 / rustfft 6.4.1 (FFT): restated as the mathematical DFT,
 *     evaluated in double precision and rounded to f32 ("ideal rustfft");
 *     pinned only by stft.rs:173-196 → beyond n_fft=4 PARITY UNPINNED
 *     (float64 numpy.fft.rfft is the cross-check, tolerance 1e-4·max|X|).
 *   - ndarray 0.17.2 dot → OpenBLAS sgemm (mel): restated as an f32 dot with
 *     sequential k order; PARITY UNPINNED (float64 matmul is the cross-check).
 *   - fast_image_resize 6.0.0 (Lanczos3 LOD): level 0 is a pinned exact copy;
 *     LOD>0 restated as textbook separable Lanczos3, PARITY UNPINNED.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------ */
/* a1. SpecSetting::calc_framing_params — spectrogram.rs:56-98         */
/* ------------------------------------------------------------------ */
static size_t next_pow2(size_t x) { /* usize::next_power_of_two */
    size_t p = 1;
    while (p < x) p <<= 1;
    return p;
}

ORC_API void orc_calc_framing_params(double win_ms, uint32_t t_overlap, uint32_t f_overlap,
                                     uint32_t sr, size_t *hop, size_t *win, size_t *n_fft) {
    double win_f = win_ms * (double)sr / 1000.0;             /* spectrogram.rs:91-93 */
    double h = round(win_f / (double)t_overlap);             /* :62-64, f64::round = half away */
    size_t hop_length = (h <= 0.0) ? 0 : (size_t)h;          /* `as usize` saturates at 0 */
    size_t win_length = hop_length * (size_t)t_overlap;      /* :86-88 */
    *hop = hop_length;
    *win = win_length;
    *n_fft = next_pow2(win_length) * (size_t)f_overlap;      /* :95-98 */
}

/* ------------------------------------------------------------------ */
/* a2. windows.rs:12-38,68-83 — periodic Hann / n_fft, all in f32      */
/* ------------------------------------------------------------------ */
ORC_API void orc_hann(size_t size, int symmetric, float *out) {
    /* cosine_window(a=0.5,b=0.5,c=0,d=0,size,symmetric) windows.rs:68-83 */
    size_t size2 = symmetric ? size : size + 1;
    const float pi = (float)M_PI; /* A::PI() for f32 */
    const float a = 0.5f, b = 0.5f, c = 0.0f, d = 0.0f;
    for (size_t i = 0; i < size; i++) {
        float x = pi * (float)i / (float)(size2 - 1);
        float b_ = b * cosf(2.0f * x);
        float c_ = c * cosf(4.0f * x);
        float d_ = d * cosf(6.0f * x);
        out[i] = (a - b_) + (c_ - d_);
    }
}

ORC_API void orc_calc_normalized_win(size_t win, size_t norm_factor, float *out) {
    /* calc_normalized_win(Hann, size, n_fft) = hann(size,false) / n_fft  windows.rs:12-28 */
    orc_hann(win, 0, out);
    float nf = (float)norm_factor;
    for (size_t i = 0; i < win; i++) out[i] = out[i] / nf;
}

/* ------------------------------------------------------------------ */
/* a3. Pad::pad — utils.rs:61-142                                      */
/* ------------------------------------------------------------------ */
/* reflect with cycling: left pad walks chain(x[1:], rev(x)[1:]).cycle()  utils.rs:111-123
 * right pad walks chain(rev(x)[1:], x[1:]).cycle()                       utils.rs:125-138 */
ORC_API void orc_pad_reflect(const float *x, size_t n, size_t left, size_t right, float *out) {
    memcpy(out + left, x, n * sizeof(float));
    if (n == 0) return;
    size_t cyc = 2 * (n - 1); /* length of the chained iterator */
    if (cyc == 0) {
        /* N == 1: the reference's cycle is empty and leaves the pad uninitialised
         * (utils.rs:91,140).  Defined here as sample replication; not a parity case. */
        for (size_t i = 0; i < left; i++) out[i] = x[0];
        for (size_t i = 0; i < right; i++) out[left + n + i] = x[0];
        return;
    }
    for (size_t i = 0; i < left; i++) {
        size_t j = i % cyc; /* i-th element of the cycled chain */
        float v = (j < n - 1) ? x[1 + j] : x[n - 2 - (j - (n - 1))];
        out[left - 1 - i] = v; /* .take(n_pad_left).rev() zipped: nearest first */
    }
    for (size_t i = 0; i < right; i++) {
        size_t j = i % cyc;
        float v = (j < n - 1) ? x[n - 2 - j] : x[1 + (j - (n - 1))];
        out[left + n + i] = v;
    }
}

ORC_API void orc_pad_constant(const float *x, size_t n, size_t left, size_t right, float c,
                              float *out) {
    for (size_t i = 0; i < left; i++) out[i] = c;
    memcpy(out + left, x, n * sizeof(float));
    for (size_t i = 0; i < right; i++) out[left + n + i] = c;
}

/* ------------------------------------------------------------------ */
/* a5. forward real FFT (realfft R2C, unnormalised) — restated as DFT  */
/* ------------------------------------------------------------------ */
static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

/* in-place iterative radix-2 complex FFT in double, sign = -1 */
static void fft_c2c_f64(double *re, double *im, size_t n) {
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) {
            double t = re[i]; re[i] = re[j]; re[j] = t;
            t = im[i]; im[i] = im[j]; im[j] = t;
        }
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        size_t half = len >> 1;
        for (size_t k = 0; k < half; k++) {
            double ang = -2.0 * M_PI * (double)k / (double)len;
            double wr = cos(ang), wi = sin(ang);
            for (size_t s = k; s < n; s += len) {
                size_t t = s + half;
                double xr = re[t] * wr - im[t] * wi;
                double xi = re[t] * wi + im[t] * wr;
                re[t] = re[s] - xr; im[t] = im[s] - xi;
                re[s] += xr;        im[s] += xi;
            }
        }
    }
}

/* X[k] = sum_n x[n] e^{-2 pi i k n / n_fft}, k = 0..n_fft/2; out interleaved (re,im) f32 */
static void rfft_f64(const float *x, size_t n_fft, float *out_ri, double *re, double *im) {
    size_t F = n_fft / 2 + 1;
    if (is_pow2(n_fft)) {
        for (size_t i = 0; i < n_fft; i++) { re[i] = (double)x[i]; im[i] = 0.0; }
        fft_c2c_f64(re, im, n_fft);
        for (size_t k = 0; k < F; k++) { out_ri[2 * k] = (float)re[k]; out_ri[2 * k + 1] = (float)im[k]; }
    } else if (n_fft % 2 == 0 && n_fft >= 64) {
        /* n_fft = M * odd, M a power of two (f_overlap = 3, 5, 6, ...: spectrogram.rs:66-72).  The same mathematical DFT in double,
         * as `odd` interleaved sub-sequences: X[k] = sum_r W_N^{r k} F_r[k mod M], F_r = FFT_M(x[r], x[r + odd], ...) — O(N log N +
         * odd N) instead of the O(N^2) of the direct sum below (which stays for small sizes and is what
         * tests/test_oracle_numpy.py checks this branch against). */
        size_t odd = n_fft, M = 1;
        while (odd % 2 == 0) { odd /= 2; M *= 2; }
        double *fr = (double *)malloc(sizeof(double) * 2 * n_fft);  /* F_r: re at [r * M ..], im at [n_fft + r * M ..] */
        for (size_t r = 0; r < odd; r++) {
            for (size_t m = 0; m < M; m++) { re[m] = (double)x[r + odd * m]; im[m] = 0.0; }
            fft_c2c_f64(re, im, M);
            memcpy(fr + r * M, re, sizeof(double) * M);
            memcpy(fr + n_fft + r * M, im, sizeof(double) * M);
        }
        for (size_t k = 0; k < F; k++) {
            double sr = 0.0, si = 0.0;
            const size_t km = k % M;
            for (size_t r = 0; r < odd; r++) {
                const double ang = -2.0 * M_PI * (double)((r * k) % n_fft) / (double)n_fft;
                const double wr = cos(ang), wi = sin(ang), ar = fr[r * M + km], ai = fr[n_fft + r * M + km];
                sr += ar * wr - ai * wi;
                si += ar * wi + ai * wr;
            }
            out_ri[2 * k] = (float)sr; out_ri[2 * k + 1] = (float)si;
        }
        free(fr);
    } else { /* direct DFT, any length (small sizes only) */
        for (size_t k = 0; k < F; k++) {
            double sr = 0.0, si = 0.0;
            for (size_t n = 0; n < n_fft; n++) {
                double ang = -2.0 * M_PI * (double)((k * n) % n_fft) / (double)n_fft;
                sr += (double)x[n] * cos(ang);
                si += (double)x[n] * sin(ang);
            }
            out_ri[2 * k] = (float)sr; out_ri[2 * k + 1] = (float)si;
        }
    }
}

/* f32 arithmetic real FFT used only for the CPU-baseline timing leg: length-n/2
 * complex radix-2 FFT on packed even/odd samples + split post-pass, twiddles
 * tabulated from double.  Same mathematical result as rfft_f64 to ~1e-6 rel. */
typedef struct { size_t n; float *twr, *twi; float *pwr, *pwi; uint32_t *rev; } orc_rfft32_plan;

ORC_API void *orc_rfft32_plan_create(size_t n_fft) {
    if (!is_pow2(n_fft) || n_fft < 4) return NULL;
    orc_rfft32_plan *p = (orc_rfft32_plan *)calloc(1, sizeof(*p));
    size_t m = n_fft / 2;
    p->n = n_fft;
    p->twr = (float *)malloc(sizeof(float) * m); p->twi = (float *)malloc(sizeof(float) * m);
    p->pwr = (float *)malloc(sizeof(float) * (m + 1)); p->pwi = (float *)malloc(sizeof(float) * (m + 1));
    p->rev = (uint32_t *)malloc(sizeof(uint32_t) * m);
    for (size_t k = 0; k < m; k++) { /* W_m^k, only k < m/2 used per stage via stride */
        double a = -2.0 * M_PI * (double)k / (double)m;
        p->twr[k] = (float)cos(a); p->twi[k] = (float)sin(a);
    }
    for (size_t k = 0; k <= m; k++) {
        double a = -2.0 * M_PI * (double)k / (double)n_fft;
        p->pwr[k] = (float)cos(a); p->pwi[k] = (float)sin(a);
    }
    unsigned lg = 0; while (((size_t)1 << lg) < m) lg++;
    for (size_t i = 0; i < m; i++) {
        uint32_t r = 0;
        for (unsigned b = 0; b < lg; b++) if (i & ((size_t)1 << b)) r |= 1u << (lg - 1 - b);
        p->rev[i] = r;
    }
    return p;
}

ORC_API void orc_rfft32_plan_destroy(void *pp) {
    orc_rfft32_plan *p = (orc_rfft32_plan *)pp;
    if (!p) return;
    free(p->twr); free(p->twi); free(p->pwr); free(p->pwi); free(p->rev); free(p);
}

/* work: 2*(n_fft/2) floats */
static void rfft_f32(const orc_rfft32_plan *p, const float *x, float *out_ri, float *work) {
    size_t m = p->n / 2;
    float *zr = work, *zi = work + m;
    for (size_t i = 0; i < m; i++) { size_t r = p->rev[i]; zr[r] = x[2 * i]; zi[r] = x[2 * i + 1]; }
    for (size_t len = 2; len <= m; len <<= 1) {
        size_t half = len >> 1, stride = m / len;
        for (size_t s = 0; s < m; s += len) {
            for (size_t k = 0; k < half; k++) {
                float wr = p->twr[k * stride], wi = p->twi[k * stride];
                size_t a = s + k, b = a + half;
                float xr = zr[b] * wr - zi[b] * wi, xi = zr[b] * wi + zi[b] * wr;
                zr[b] = zr[a] - xr; zi[b] = zi[a] - xi;
                zr[a] += xr;        zi[a] += xi;
            }
        }
    }
    for (size_t k = 0; k <= m; k++) {
        size_t k1 = k % m, k2 = (m - k) % m;
        float ar = zr[k1], ai = zi[k1], br = zr[k2], bi = -zi[k2]; /* conj(Z[m-k]) */
        float er = 0.5f * (ar + br), ei = 0.5f * (ai + bi);        /* even part  */
        float dr = 0.5f * (ar - br), di = 0.5f * (ai - bi);        /* (Z - conj Z')/2 */
        /* odd part = -i * d ; X = e + W^k * odd */
        float orr = di, oi = -dr;
        float wr = p->pwr[k], wi = p->pwi[k];
        out_ri[2 * k] = er + (orr * wr - oi * wi);
        out_ri[2 * k + 1] = ei + (orr * wi + oi * wr);
    }
}

/* ------------------------------------------------------------------ */
/* a4. perform_stft + to_windowed_frames — stft.rs:16-149              */
/* ------------------------------------------------------------------ */
/* ndarray windows_with_stride(win, hop).into_iter().count() */
static size_t n_windows(size_t len, size_t win, size_t hop) {
    if (len < win || win == 0) return 0;
    return (len - win) / hop + 1;
}

typedef struct { const float *window; size_t win, hop, n_fft, pl; int fft32; const orc_rfft32_plan *plan;
                 float *frame; double *re, *im; float *work; } stft_ctx;

/* to_windowed_frames + do_fft over one contiguous (already padded) segment  stft.rs:127-149,44-48 */
static size_t stft_segment(const stft_ctx *c, const float *seg, size_t seg_len, float *out_ri) {
    size_t F = c->n_fft / 2 + 1;
    size_t nfr = n_windows(seg_len, c->win, c->hop);
    for (size_t f = 0; f < nfr; f++) {
        const float *x = seg + f * c->hop;
        memset(c->frame, 0, sizeof(float) * c->n_fft);          /* zero pads  stft.rs:139,144 */
        for (size_t i = 0; i < c->win; i++) c->frame[c->pl + i] = x[i] * c->window[i]; /* :140-143 */
        if (c->fft32) rfft_f32(c->plan, c->frame, out_ri + 2 * F * f, c->work);
        else rfft_f64(c->frame, c->n_fft, out_ri + 2 * F * f, c->re, c->im);
    }
    return nfr;
}

/* number of frames perform_stft returns (literal walk of stft.rs:50-97) */
ORC_API size_t orc_stft_n_frames(size_t n, size_t win, size_t hop) {
    if (n == 0 || win == 0 || hop == 0) return 0;
    if (n < win) return n_windows(n + 2 * (win / 2), win, hop);           /* :50-54 */
    size_t n_front = n_windows(win - 1 + win / 2, win, hop);              /* :77-81 */
    size_t first_i = n_front * hop - win / 2;                             /* :83 */
    size_t n_mid = n_windows(n - first_i, win, hop);                      /* :84-85 */
    first_i += n_mid * hop;                                               /* :87 */
    size_t i_back = first_i < n - win / 2 - 1 ? first_i : n - win / 2 - 1; /* :88 */
    size_t back_len = n - i_back + win / 2;                               /* :90-93 */
    size_t skip = first_i > i_back ? first_i - i_back : 0;                /* :94 */
    size_t n_back = skip <= back_len ? n_windows(back_len - skip, win, hop) : 0;
    return n_front + n_mid + n_back;
}

/* out_ri: T x F complex interleaved.  window = normalised window (len win).
 * fft32 != 0 selects the f32 FFT (baseline timing only). Returns T. */
ORC_API size_t orc_perform_stft(const float *x, size_t n, size_t win, size_t hop, size_t n_fft,
                                const float *window, int fft32, float *out_ri) {
    if (n == 0 || win == 0 || hop == 0 || n_fft < win) return 0;
    size_t F = n_fft / 2 + 1;
    stft_ctx c; memset(&c, 0, sizeof c);
    c.window = window; c.win = win; c.hop = hop; c.n_fft = n_fft;
    c.pl = (n_fft - win) / 2;                                  /* n_pad_left stft.rs:36 */
    c.frame = (float *)malloc(sizeof(float) * n_fft);
    orc_rfft32_plan *plan = NULL;
    if (fft32 && is_pow2(n_fft) && n_fft >= 4) {
        plan = (orc_rfft32_plan *)orc_rfft32_plan_create(n_fft);
        c.fft32 = 1; c.plan = plan; c.work = (float *)malloc(sizeof(float) * n_fft);
    } else {
        c.re = (double *)malloc(sizeof(double) * n_fft); c.im = (double *)malloc(sizeof(double) * n_fft);
    }
    size_t T = 0;
    if (n < win) {                                             /* stft.rs:50-76 */
        size_t pl = win / 2, plen = n + 2 * pl;
        float *padded = (float *)malloc(sizeof(float) * plen);
        orc_pad_reflect(x, n, pl, pl, padded);
        T = stft_segment(&c, padded, plen, out_ri);
        free(padded);
    } else {
        /* front: input[..win-1] left-reflect-padded by win/2   stft.rs:77-81 */
        size_t fl = win - 1 + win / 2;
        float *front = (float *)malloc(sizeof(float) * (fl ? fl : 1));
        orc_pad_reflect(x, win - 1, win / 2, 0, front);
        size_t n_front = stft_segment(&c, front, fl, out_ri);
        free(front);
        size_t first_i = n_front * hop - win / 2;               /* :83 */
        size_t n_mid = stft_segment(&c, x + first_i, n - first_i, out_ri + 2 * F * n_front); /* :84-85 */
        first_i += n_mid * hop;                                 /* :87 */
        size_t i_back = first_i < n - win / 2 - 1 ? first_i : n - win / 2 - 1; /* :88 */
        size_t bl = n - i_back + win / 2;
        float *back = (float *)malloc(sizeof(float) * bl);
        orc_pad_reflect(x + i_back, n - i_back, 0, win / 2, back); /* :90-93 */
        size_t skip = first_i > i_back ? first_i - i_back : 0;  /* :94 */
        size_t n_back = 0;
        if (skip <= bl) n_back = stft_segment(&c, back + skip, bl - skip, out_ri + 2 * F * (n_front + n_mid));
        free(back);
        T = n_front + n_mid + n_back;
    }
    free(c.frame); free(c.re); free(c.im); free(c.work);
    orc_rfft32_plan_destroy(plan);
    return T;
}

/* ------------------------------------------------------------------ */
/* a9. dB_from_amp_inplace_default — decibel.rs:170-214                */
/* ------------------------------------------------------------------ */
ORC_API void orc_dB_from_amp_inplace(float *x, size_t n, float ref_value, float amin) {
    /* log_for_dB_inplace decibel.rs:170-194 */
    if (isnan(ref_value)) return;
    if (signbit(ref_value)) { for (size_t i = 0; i < n; i++) x[i] = NAN; return; }
    float log_amin = log10f(amin);
    float log_ref = (ref_value > amin) ? log10f(ref_value) : log_amin;
    float out_for_small = log_amin - log_ref;
    for (size_t i = 0; i < n; i++) {
        float v = x[i];
        if (isnan(v) || signbit(v)) x[i] = NAN;
        else if (v > amin) x[i] = log10f(v) - log_ref;
        else x[i] = out_for_small;
    }
    /* scalar_mul_simd_inplace(20) decibel.rs:198-202, simd.rs:185-207 */
    for (size_t i = 0; i < n; i++) x[i] *= 20.0f;
}

/* ------------------------------------------------------------------ */
/* a7. mel scale + filterbank — src-common/src/lib.rs:11-103           */
/* ------------------------------------------------------------------ */
#define MIN_LOG_MEL 15
#define MIN_LOG_HZ 1000.0
#define LOGSTEP 0.06875177742094912
#define LINEARSCALE (200.0 / 3.0)

ORC_API double orc_mel_to_hz_f64(double mel) { /* lib.rs:17-29 */
    if (mel < (double)MIN_LOG_MEL) return LINEARSCALE * mel;
    return MIN_LOG_HZ * exp(LOGSTEP * (mel - (double)MIN_LOG_MEL));
}
ORC_API double orc_mel_from_hz_f64(double hz) { /* lib.rs:31-43 */
    if (hz < MIN_LOG_HZ) return hz / LINEARSCALE;
    return (double)MIN_LOG_MEL + log(hz / MIN_LOG_HZ) / LOGSTEP;
}
ORC_API float orc_mel_to_hz_f32(float mel) {
    if (mel < (float)MIN_LOG_MEL) return (float)LINEARSCALE * mel;
    return (float)MIN_LOG_HZ * expf((float)LOGSTEP * (mel - (float)MIN_LOG_MEL));
}
ORC_API float orc_mel_from_hz_f32(float hz) {
    if (hz < (float)MIN_LOG_HZ) return hz / (float)LINEARSCALE;
    return (float)MIN_LOG_MEL + logf(hz / (float)MIN_LOG_HZ) / (float)LOGSTEP;
}

/* ndarray Array::linspace(a, b, n): step = (b-a)/(n-1), x_i = a + step*i */
static void linspace_f32(float a, float b, size_t n, float *out) {
    float step = (n > 1) ? (b - a) / (float)(n - 1) : 0.0f;
    for (size_t i = 0; i < n; i++) out[i] = a + step * (float)i;
}
static void linspace_f64(double a, double b, size_t n, double *out) {
    double step = (n > 1) ? (b - a) / (double)(n - 1) : 0.0;
    for (size_t i = 0; i < n; i++) out[i] = a + step * (double)i;
}

/* The two f32 frequency arrays calc_mel_fb works from (lib.rs:61-67): lin[n_fft/2+1] = the bins' frequencies,
 * mf[n_mel+2] = the triangle points.  For tests of table builders that need the filters' geometry, not only their values. */
ORC_API void orc_mel_fb_points_f32(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, float *lin, float *mf) {
    float f_nyquist = (float)(((double)sr) / 2.0);
    if (fmax < 0.0f) fmax = f_nyquist;
    linspace_f32(0.0f, f_nyquist, n_fft / 2 + 1, lin);
    linspace_f32(orc_mel_from_hz_f32(fmin), orc_mel_from_hz_f32(fmax), n_mel + 2, mf);
    for (size_t i = 0; i < n_mel + 2; i++) mf[i] = orc_mel_to_hz_f32(mf[i]);
}

/* calc_mel_fb::<f32>  lib.rs:46-89.  out: (n_fft/2+1) x n_mel, C order.
 * fmax < 0 means None (nyquist). */
ORC_API void orc_calc_mel_fb_f32(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax,
                                 int do_norm, float *out) {
    float f_nyquist = (float)(((double)sr) / 2.0);
    if (fmax < 0.0f) fmax = f_nyquist;
    size_t n_freq = n_fft / 2 + 1;
    float *lin = (float *)malloc(sizeof(float) * n_freq);
    float *mf = (float *)malloc(sizeof(float) * (n_mel + 2));
    float *w = (float *)malloc(sizeof(float) * n_freq);
    linspace_f32(0.0f, f_nyquist, n_freq, lin);
    linspace_f32(orc_mel_from_hz_f32(fmin), orc_mel_from_hz_f32(fmax), n_mel + 2, mf);
    for (size_t i = 0; i < n_mel + 2; i++) mf[i] = orc_mel_to_hz_f32(mf[i]);
    for (size_t m = 0; m < n_mel; m++) {
        memset(w, 0, sizeof(float) * n_freq);
        for (size_t i = 0; i < n_freq; i++) {
            float f = lin[i];
            if (f <= mf[m]) continue;
            else if (mf[m] < f && f < mf[m + 1]) w[i] = (f - mf[m]) / (mf[m + 1] - mf[m]);
            else if (f == mf[m + 1]) w[i] = 1.0f;
            else if (mf[m + 1] < f && f < mf[m + 2]) w[i] = (mf[m + 2] - f) / (mf[m + 2] - mf[m + 1]);
            else break;
        }
        if (do_norm) {
            float s = 0.0f; /* ndarray sum(): restated sequentially (filters have <70 non-zeros) */
            for (size_t i = 0; i < n_freq; i++) s += w[i];
            float dv = s > 1.1920929e-07f ? s : 1.1920929e-07f; /* .max(f32::EPSILON) */
            for (size_t i = 0; i < n_freq; i++) w[i] /= dv;
        }
        for (size_t i = 0; i < n_freq; i++) out[i * n_mel + m] = w[i]; /* weights.t() lib.rs:88 */
    }
    free(lin); free(mf); free(w);
}

ORC_API void orc_calc_mel_fb_f64(uint32_t sr, size_t n_fft, size_t n_mel, double fmin, double fmax,
                                 int do_norm, double *out) {
    double f_nyquist = ((double)sr) / 2.0;
    if (fmax < 0.0) fmax = f_nyquist;
    size_t n_freq = n_fft / 2 + 1;
    double *lin = (double *)malloc(sizeof(double) * n_freq);
    double *mf = (double *)malloc(sizeof(double) * (n_mel + 2));
    double *w = (double *)malloc(sizeof(double) * n_freq);
    linspace_f64(0.0, f_nyquist, n_freq, lin);
    linspace_f64(orc_mel_from_hz_f64(fmin), orc_mel_from_hz_f64(fmax), n_mel + 2, mf);
    for (size_t i = 0; i < n_mel + 2; i++) mf[i] = orc_mel_to_hz_f64(mf[i]);
    for (size_t m = 0; m < n_mel; m++) {
        memset(w, 0, sizeof(double) * n_freq);
        for (size_t i = 0; i < n_freq; i++) {
            double f = lin[i];
            if (f <= mf[m]) continue;
            else if (mf[m] < f && f < mf[m + 1]) w[i] = (f - mf[m]) / (mf[m + 1] - mf[m]);
            else if (f == mf[m + 1]) w[i] = 1.0;
            else if (mf[m + 1] < f && f < mf[m + 2]) w[i] = (mf[m + 2] - f) / (mf[m + 2] - mf[m + 1]);
            else break;
        }
        if (do_norm) {
            double s = 0.0;
            for (size_t i = 0; i < n_freq; i++) s += w[i];
            double dv = s > 2.220446049250313e-16 ? s : 2.220446049250313e-16;
            for (size_t i = 0; i < n_freq; i++) w[i] /= dv;
        }
        for (size_t i = 0; i < n_freq; i++) out[i * n_mel + m] = w[i];
    }
    free(lin); free(mf); free(w);
}

/* calc_mel_fb_default's n_mel search — lib.rs:91-103.  Returns n_mel. */
ORC_API size_t orc_mel_default_n_mel(uint32_t sr, size_t n_fft) {
    float r = orc_mel_from_hz_f32((float)sr / 2.0f) / orc_mel_from_hz_f32((float)sr / (float)n_fft);
    float v = fmaf(r, 2.0f, -1.0f);                       /* .mul_add(2., -1.) */
    size_t n_mel = (v <= 0.0f || isnan(v)) ? 0 : (size_t)v; /* `as usize` */
    size_t n_freq = n_fft / 2 + 1;
    if (n_mel > n_freq) n_mel = n_freq;
    float *fb = (float *)malloc(sizeof(float) * n_freq * (n_mel ? n_mel : 1));
    while (n_mel > 0) {
        orc_calc_mel_fb_f32(sr, n_fft, n_mel, 0.0f, -1.0f, 1, fb);
        int ok = 1;
        for (size_t m = 0; m < n_mel && ok; m++) { /* sum_axis(Axis(0)) > 0 for all */
            float s = 0.0f;
            for (size_t i = 0; i < n_freq; i++) s += fb[i * n_mel + m];
            if (!(s > 0.0f)) ok = 0;
        }
        if (ok) break;
        n_mel--;
    }
    free(fb);
    return n_mel;
}

/* FreqScale::hz_range_to_idx — lib.rs:134-159.  freq_scale: 0 linear, 1 mel */
ORC_API void orc_hz_range_to_idx(int freq_scale, float hz0, float hz1, uint32_t sr, size_t n,
                                 size_t *i0, size_t *i1) {
    if (hz0 >= hz1) { *i0 = 0; *i1 = 0; return; }
    float half_sr = (float)sr / 2.0f;
    float r0 = freq_scale ? orc_mel_from_hz_f32(hz0) / orc_mel_from_hz_f32(half_sr) : hz0 / half_sr;
    float r1 = freq_scale ? orc_mel_from_hz_f32(hz1) / orc_mel_from_hz_f32(half_sr) : hz1 / half_sr;
    float lo = floorf(r0 * (float)n); if (!(lo > 0.0f)) lo = 0.0f;
    float hi = ceilf(r1 * (float)n);
    *i0 = (size_t)lo;
    *i1 = hi <= 0.0f ? 0 : (size_t)hi;
}

/* ------------------------------------------------------------------ */
/* calc_spec — spectrogram.rs:187-212                                  */
/* ------------------------------------------------------------------ */
/* out: T x H f32 dB, H = n_fft/2+1 (linear) or n_mel.  mel_fb: F x n_mel or NULL.
 * If amp_out != NULL it receives the T x F linear amplitude (|X|) before mel/dB. */
ORC_API size_t orc_calc_spec(const float *x, size_t n, size_t win, size_t hop, size_t n_fft,
                             const float *window, const float *mel_fb, size_t n_mel, int fft32,
                             float *out, float *amp_out) {
    size_t F = n_fft / 2 + 1;
    size_t T = orc_stft_n_frames(n, win, hop);
    if (T == 0) return 0;
    float *ri = (float *)malloc(sizeof(float) * 2 * F * T);
    size_t T2 = orc_perform_stft(x, n, win, hop, n_fft, window, fft32, ri);
    if (T2 != T) { free(ri); return 0; }
    float *lin = mel_fb ? (float *)malloc(sizeof(float) * F * T) : out;
    for (size_t i = 0; i < T * F; i++) lin[i] = hypotf(ri[2 * i], ri[2 * i + 1]); /* Complex::norm :200 */
    free(ri);
    if (amp_out) memcpy(amp_out, lin, sizeof(float) * F * T);
    if (mel_fb) { /* linspec.dot(&mel_fb) :207 (sgemm restated, k sequential) */
        for (size_t t = 0; t < T; t++) {
            float *o = out + t * n_mel;
            for (size_t m = 0; m < n_mel; m++) o[m] = 0.0f;
            for (size_t f = 0; f < F; f++) {
                float a = lin[t * F + f];
                const float *row = mel_fb + f * n_mel;
                for (size_t m = 0; m < n_mel; m++) o[m] += a * row[m];
            }
        }
        free(lin);
        orc_dB_from_amp_inplace(out, T * n_mel, 1.0f, 0.0f); /* :208 */
    } else {
        orc_dB_from_amp_inplace(out, T * F, 1.0f, 0.0f);     /* :203 */
    }
    return T;
}

/* ------------------------------------------------------------------ */
/* a10/a6. simd.rs reductions (AVX2 tier order; scalar semantics)      */
/* ------------------------------------------------------------------ */
static float rs_min(float a, float b) { /* f32::min: NaN-ignoring */
    if (isnan(a)) return b; if (isnan(b)) return a; return a < b ? a : b;
}
static float rs_max(float a, float b) {
    if (isnan(a)) return b; if (isnan(b)) return a; return a > b ? a : b;
}

ORC_API void orc_find_min_max(const float *x, size_t n, float *mn, float *mx) {
    /* simd.rs:271-303: start at (+inf,-inf); empty → unchanged. Order-independent w/o NaN. */
    float lo = INFINITY, hi = -INFINITY;
    for (size_t i = 0; i < n; i++) { lo = rs_min(lo, x[i]); hi = rs_max(hi, x[i]); }
    *mn = lo; *mx = hi;
}

/* simd::sum, AVX2 tier (simd.rs:591-619, 808-818): scalar prefix up to 32-byte
 * alignment (`misalign` = number of prefix elements, 0..7), 8 lane accumulators,
 * reduce (lo+hi), (s0+s1)+(s2+s3), scalar suffix. */
ORC_API float orc_sum_avx2(const float *x, size_t n, size_t misalign) {
    if (n == 0) return 0.0f;
    float sum = 0.0f;
    size_t pre = misalign < n ? misalign : n;
    size_t mid = (n - pre) / 8;
    if (mid == 0) { pre = n; } /* align_to yields everything in prefix when no full chunk */
    size_t i = 0;
    for (; i < pre; i++) sum += x[i];
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    size_t nm = (n - pre) / 8;
    for (size_t c = 0; c < nm; c++, i += 8)
        for (int l = 0; l < 8; l++) v[l] += x[i + l];
    float s0 = v[0] + v[4], s1 = v[1] + v[5], s2 = v[2] + v[6], s3 = v[3] + v[7];
    sum += (s0 + s1) + (s2 + s3);
    for (; i < n; i++) sum += x[i];
    return sum;
}

/* simd::sum_squares, scalar tier (simd.rs:820-832): Kahan-compensated f32 sum of x*x.  (The SIMD tiers keep
 * per-lane compensations, :705-750; all agree to ~1 ulp of the exact sum.) */
ORC_API float orc_sum_squares(const float *x, size_t n) {
    volatile float sum = 0.0f, c = 0.0f; /* volatile: keep the compensation from being optimised away */
    for (size_t i = 0; i < n; i++) {
        volatile float y = x[i] * x[i] - c;
        volatile float t = sum + y;
        c = (t - sum) - y;
        sum = t;
    }
    return sum;
}

/* simd::abs_max, scalar tier (simd.rs:935-937): fold(0.0, max(|x|)); f32::max ignores NaN */
ORC_API float orc_abs_max(const float *x, size_t n) {
    float m = 0.0f;
    for (size_t i = 0; i < n; i++) m = rs_max(m, fabsf(x[i]));
    return m;
}

ORC_API void orc_scalar_mul(float *x, size_t n, float s) { /* simd.rs:185-207 */
    for (size_t i = 0; i < n; i++) x[i] *= s;
}

/* update_spec_imgs' range clamp — core/mod.rs:169-180 */
ORC_API void orc_global_db_range(const float *mins, const float *maxs, size_t n, float dB_range,
                                 float *min_dB, float *max_dB) {
    float mn = INFINITY, mx = -INFINITY;
    for (size_t i = 0; i < n; i++) { mn = rs_min(mn, mins[i]); mx = rs_max(mx, maxs[i]); }
    mx = rs_min(mx, 0.0f);
    mn = rs_max(mn, mx - dB_range);
    *min_dB = mn; *max_dB = mx;
}

/* ------------------------------------------------------------------ */
/* a12. convert_spectrogram_to_img — visualize/drawing.rs:4-33         */
/* ------------------------------------------------------------------ */
static uint16_t sat_u16(float v) { /* Rust `as u16`: saturating, NaN → 0 */
    if (isnan(v)) return 0;
    if (v <= 0.0f) return 0;
    if (v >= 65535.0f) return 65535;
    return (uint16_t)v;
}

/* spec: T x H f32.  out: (i1-i0) x T u16.  colormap_len == 0 means None. */
ORC_API void orc_convert_spectrogram_to_img(const float *spec, size_t T, size_t H, size_t i0,
                                            size_t i1, float min_dB, float max_dB,
                                            uint32_t colormap_len, uint16_t *out) {
    size_t height = i1 - i0, width = T;
    float span = max_dB - min_dB;
    if (min_dB == max_dB && max_dB == -INFINITY) {            /* :16-18 */
        memset(out, 0, sizeof(uint16_t) * height * width);
        return;
    }
    uint16_t min_value = 1;                                    /* :20-21 */
    if (colormap_len) {
        double r = round(65535.0 / (double)colormap_len);
        uint16_t v = r >= 65535.0 ? 65535 : (r <= 0.0 ? 0 : (uint16_t)r);
        min_value = v > 1 ? v : 1;
    }
    float u16_span = (float)(65535 - min_value);               /* :22 */
    for (size_t i = 0; i < height; i++) {
        size_t i_freq = i0 + i;
        for (size_t j = 0; j < width; j++) {
            uint16_t px = 0;
            if (i_freq < H) {                                   /* :25 */
                float z = (spec[j * H + i_freq] - min_dB) / span;
                float u = z * u16_span + (float)min_value;
                float r = roundf(u);                            /* f32::round half away */
                if (r < 0.0f) r = 0.0f; else if (r > 65535.0f) r = 65535.0f; /* clamp; NaN stays NaN */
                px = sat_u16(r);
            }
            out[i * width + j] = px;
        }
    }
}

/* ------------------------------------------------------------------ */
/* a14. encode_waveform_tile — render_tiles.rs:232-279                 */
/* ------------------------------------------------------------------ */
static void put_u32(uint8_t *p, uint32_t v) { p[0] = v; p[1] = v >> 8; p[2] = v >> 16; p[3] = v >> 24; }
static void put_u64(uint8_t *p, uint64_t v) { for (int i = 0; i < 8; i++) p[i] = (uint8_t)(v >> (8 * i)); }
static void put_f32(uint8_t *p, float f) { uint32_t u; memcpy(&u, &f, 4); put_u32(p, u); }

static size_t sat_mul(size_t a, size_t b) { size_t r; return __builtin_mul_overflow(a, b, &r) ? SIZE_MAX : r; }
static size_t sat_add(size_t a, size_t b) { size_t r; return __builtin_add_overflow(a, b, &r) ? SIZE_MAX : r; }

static void waveform_bin_stats(const float *s, size_t n, float *mn, float *mx, float *rep) {
    if (n >= 32) {                                             /* :264-268 (SIMD tier) */
        orc_find_min_max(s, n, mn, mx);
        size_t mis = (size_t)((32 - ((uintptr_t)s & 31)) & 31) / 4; /* align_to::<__m256> prefix */
        *rep = orc_sum_avx2(s, n, mis) / (float)n;
        return;
    }
    float lo = INFINITY, hi = -INFINITY, sum = 0.0f;           /* :270-278 */
    for (size_t i = 0; i < n; i++) { lo = rs_min(lo, s[i]); hi = rs_max(hi, s[i]); sum += s[i]; }
    *mn = lo; *mx = hi; *rep = sum / (float)n;
}

/* returns bytes written (24 + bins*12); out must hold 24 + 1024*12 */
ORC_API size_t orc_encode_waveform_tile(const float *wav, size_t n, uint64_t revision, uint32_t level,
                                        uint32_t tile_index, uint8_t *out) {
    size_t spb = level < 64 ? ((size_t)1 << level) : SIZE_MAX;  /* checked_shl */
    if (level < 64 && (spb >> level) != 1) spb = SIZE_MAX;
    size_t tile_samples = sat_mul(1024, spb);
    size_t start = sat_mul((size_t)tile_index, tile_samples);
    size_t end = sat_add(start, tile_samples); if (end > n) end = n;
    size_t bins = start >= end ? 0 : (end - start) / spb + ((end - start) % spb != 0); /* div_ceil */
    put_u64(out, revision);
    put_u32(out + 8, (uint32_t)bins);
    put_u32(out + 12, spb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)spb);
    put_u32(out + 16, tile_index);
    put_u32(out + 20, 0);
    for (size_t b = 0; b < bins; b++) {
        size_t bs = start + b * spb;
        size_t be = sat_add(bs, spb); if (be > end) be = end;
        float mn, mx, rep;
        waveform_bin_stats(wav + bs, be - bs, &mn, &mx, &rep);
        put_f32(out + 24 + 12 * b, mn); put_f32(out + 28 + 12 * b, mx); put_f32(out + 32 + 12 * b, rep);
    }
    return 24 + bins * 12;
}

/* ------------------------------------------------------------------ */
/* a13. encode_spectrogram_tile — render_tiles.rs:281-393              */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t width, height, origin_x, origin_y; size_t lod_w, lod_h; } orc_tile_geom;

/* geometry only (render_tiles.rs:290-313); W = img cols (time), Hh = img rows (freq) */
ORC_API void orc_spectrogram_tile_geom(size_t W, size_t Hh, uint32_t level_x, uint32_t level_y,
                                       uint32_t tile_x, uint32_t tile_y, orc_tile_geom *g) {
    size_t sx = level_x < 64 ? ((size_t)1 << level_x) : SIZE_MAX;
    size_t sy = level_y < 64 ? ((size_t)1 << level_y) : SIZE_MAX;
    size_t lod_w = W / sx + (W % sx != 0), lod_h = Hh / sy + (Hh % sy != 0);
    size_t start_x = sat_mul(tile_x, 512), start_y = sat_mul(tile_y, 512);
    size_t core_w = lod_w > start_x ? lod_w - start_x : 0; if (core_w > 512) core_w = 512;
    size_t core_h = lod_h > start_y ? lod_h - start_y : 0; if (core_h > 512) core_h = 512;
    size_t ox = start_x > 4 ? start_x - 4 : 0, oy = start_y > 4 ? start_y - 4 : 0;
    size_t w = 0, h = 0;
    if (core_w && core_h) {
        size_t rx = start_x + core_w + 4; if (rx > lod_w) rx = lod_w;
        size_t ry = start_y + core_h + 4; if (ry > lod_h) ry = lod_h;
        w = rx > ox ? rx - ox : 0; h = ry > oy ? ry - oy : 0;
    }
    g->width = (uint32_t)w; g->height = (uint32_t)h; g->origin_x = (uint32_t)ox; g->origin_y = (uint32_t)oy;
    g->lod_w = lod_w; g->lod_h = lod_h;
}

static double lanczos3(double x) {
    if (x == 0.0) return 1.0;
    if (x <= -3.0 || x >= 3.0) return 0.0;
    double px = M_PI * x;
    return 3.0 * sin(px) * sin(px / 3.0) / (px * px);
}

/* Separable Lanczos3 resample of the crop box [left, left+cw) x [top, top+ch) of a u16 image into dw x dh.
 * PARITY UNPINNED vs fast_image_resize 6.0.0 (crate absent, see header) — but since round 4 restated in the exact
 * arithmetic of the convolution that crate documents itself as following, Pillow's ImagingResample (src/libImaging/
 * Resample.c: precompute_coeffs + the 16-bit horizontal / vertical passes), and pinned to Pillow 12.2's own output
 * (tests/golden/lod_pillow_cases.npz, scripts/make_golden_lod.py):
 *   scale = extent / out_size, filterscale = max(scale, 1), support = 3 * filterscale, ss = 1 / filterscale;
 *   center = in0 + (o + 0.5) * scale;  window [ (int)(center - support + 0.5), (int)(center + support + 0.5) ) clipped at
 *   the IMAGE (never at the crop box);  tap k = lanczos(((k + xmin) - center + 0.5) * ss) with lanczos(t) =
 *   sinc(t) * sinc(t / 3) on -3 <= t < 3, sinc(t) = sin(pi t) / (pi t);  taps divided by their sum BEFORE the pass;
 *   a pixel = sum of pixel * tap in ascending order from 0.0 (multiply, then add), + 0.5, truncated; horizontal pass over
 *   the rows the vertical pass needs, rounded to u16, then the vertical pass.
 * One deliberate difference: a sum above 65535 is clamped to 65535 (u16 saturation, as a u16 resizer must); Pillow's
 * 16-bit path writes CLIP8(v >> 8) and CLIP8(v % 256) separately there, i.e. 0xFF00 | (v & 0xFF) — the fixtures' tests
 * treat exactly those pixels (and what the second pass derives from them) as Pillow's overflow artefact. */
static double pil_sinc(double x) {
    if (x == 0.0) return 1.0;
    x = x * M_PI;
    return sin(x) / x;
}
static double pil_lanczos(double x) {
    if (-3.0 <= x && x < 3.0) return pil_sinc(x) * pil_sinc(x / 3);
    return 0.0;
}
typedef struct { int *xmin, *n; double *k; int ksize; } pil_axis;
static void pil_axis_build(int in_size, double in0, double in1, int out_size, pil_axis *a) {
    double scale = (in1 - in0) / out_size, filterscale = scale;
    if (filterscale < 1.0) filterscale = 1.0;
    const double support = 3.0 * filterscale;
    a->ksize = (int)ceil(support) * 2 + 1;
    a->xmin = (int *)malloc(sizeof(int) * (size_t)(out_size ? out_size : 1));
    a->n = (int *)malloc(sizeof(int) * (size_t)(out_size ? out_size : 1));
    a->k = (double *)calloc((size_t)a->ksize * (size_t)(out_size ? out_size : 1), sizeof(double));
    for (int xx = 0; xx < out_size; xx++) {
        const double center = in0 + (xx + 0.5) * scale, ss = 1.0 / filterscale;
        double ww = 0.0;
        int xmin = (int)(center - support + 0.5);
        if (xmin < 0) xmin = 0;
        int xmax = (int)(center + support + 0.5);
        if (xmax > in_size) xmax = in_size;
        xmax -= xmin;
        if (xmax < 0) xmax = 0;
        double *k = a->k + (size_t)xx * (size_t)a->ksize;
        for (int x = 0; x < xmax; x++) {
            const double w = pil_lanczos((x + xmin - center + 0.5) * ss);
            k[x] = w;
            ww += w;
        }
        for (int x = 0; x < xmax; x++)
            if (ww != 0.0) k[x] /= ww;
        a->xmin[xx] = xmin;
        a->n[xx] = xmax;
    }
}
static uint16_t pil_round_u16(double ss) {
    const long v = (long)(ss >= 0.0 ? ss + 0.5 : ss - 0.5);  /* ROUND_UP */
    return (uint16_t)(v < 0 ? 0 : v > 65535 ? 65535 : v);
}
static void resize_lanczos3_u16(const uint16_t *src, size_t sw, size_t sh, double left, double top,
                                double cw, double ch, size_t dw, size_t dh, uint16_t *dst) {
    pil_axis ax, ay;
    pil_axis_build((int)sw, left, left + cw, (int)dw, &ax);
    pil_axis_build((int)sh, top, top + ch, (int)dh, &ay);
    /* rows needed by the vertical pass: first window's start .. last window's end (Resample.c: ybox_first / ybox_last) */
    long y_lo = dh ? ay.xmin[0] : 0, y_hi = dh ? ay.xmin[dh - 1] + ay.n[dh - 1] : 0;
    if (y_hi < y_lo) y_hi = y_lo;
    const size_t nrows = (size_t)(y_hi - y_lo);
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (nrows ? nrows : 1) * (dw ? dw : 1));
    for (size_t r = 0; r < nrows; r++) {
        const uint16_t *row = src + ((size_t)y_lo + r) * sw;
        for (size_t ox = 0; ox < dw; ox++) {
            const double *k = ax.k + ox * (size_t)ax.ksize;
            double ss = 0.0;
            for (int x = 0; x < ax.n[ox]; x++) ss += (double)row[ax.xmin[ox] + x] * k[x];
            tmp[r * dw + ox] = pil_round_u16(ss);
        }
    }
    for (size_t oy = 0; oy < dh; oy++) {
        const double *k = ay.k + oy * (size_t)ay.ksize;
        for (size_t ox = 0; ox < dw; ox++) {
            double ss = 0.0;
            for (int y = 0; y < ay.n[oy]; y++) ss += (double)tmp[(size_t)(ay.xmin[oy] + y - y_lo) * dw + ox] * k[y];
            dst[oy * dw + ox] = pil_round_u16(ss);
        }
    }
    free(tmp);
    free(ax.xmin); free(ax.n); free(ax.k);
    free(ay.xmin); free(ay.n); free(ay.k);
}

/* ------------------------------------------------------------------ */
/* Sensitivity variants of the LOD resize (NOT parity: the crate source is absent, see the header).     */
/* fast_image_resize resamples U16 images in fixed point (as Pillow does for 8-bit ones): per axis the   */
/* f64 taps are normalised, scaled by 2^precision, rounded to i32, accumulated in i64 from 2^(p-1)       */
/* (round half up) and shifted back.  mode 1 restates that with the "largest precision whose biggest     */
/* coefficient still fits an i32" rule; mode 2 with a crude fixed precision of 16 bits.  They exist to   */
/* bound how far ANY fixed-point Lanczos3 of this structure can sit from the f64 one on the colour-index */
/* plane (tests/test_oracle_lod.py), which is what "parity unpinned" leaves open.                        */
/* ------------------------------------------------------------------ */
typedef struct { long *start, *count; double *w; size_t window; int precision; int64_t *wi; } orc_axis;

static void axis_free(orc_axis *a) { free(a->start); free(a->count); free(a->w); free(a->wi); }

/* per-axis coefficient table: in_size source pixels, crop [in0, in0 + extent), out_size outputs */
static void axis_build(size_t in_size, double in0, double extent, size_t out_size, int fixed_bits, orc_axis *a) {
    double scale = extent / (double)out_size;
    double fscale = scale < 1.0 ? 1.0 : scale;
    double radius = 3.0 * fscale;
    size_t window = (size_t)ceil(radius) * 2 + 2;
    a->window = window;
    a->start = (long *)malloc(sizeof(long) * out_size);
    a->count = (long *)malloc(sizeof(long) * out_size);
    a->w = (double *)calloc(window * out_size, sizeof(double));
    a->wi = NULL; a->precision = 0;
    double wmax = 0.0;
    for (size_t o = 0; o < out_size; o++) {
        double center = in0 + ((double)o + 0.5) * scale;
        long x0 = (long)floor(center - radius), x1 = (long)ceil(center + radius);
        if (x0 < 0) x0 = 0; if (x1 > (long)in_size) x1 = (long)in_size;
        if (x1 < x0) x1 = x0;
        double ww = 0.0;
        for (long x = x0; x < x1; x++) { double w = lanczos3(((double)x + 0.5 - center) / fscale); a->w[o * window + (size_t)(x - x0)] = w; ww += w; }
        if (ww != 0.0) for (long x = x0; x < x1; x++) a->w[o * window + (size_t)(x - x0)] /= ww;
        for (long x = x0; x < x1; x++) if (a->w[o * window + (size_t)(x - x0)] > wmax) wmax = a->w[o * window + (size_t)(x - x0)];
        a->start[o] = x0; a->count[o] = x1 - x0;
    }
    if (fixed_bits != 0) {
        int p = 0;
        if (fixed_bits < 0) {  /* largest precision such that the biggest coefficient still fits an i32 */
            for (int cur = 0; cur < 50; cur++) {
                p = cur;
                double next = floor(wmax * ldexp(1.0, cur + 1) + 0.5);
                if (next >= 2147483647.0) break;
            }
        } else p = fixed_bits;
        a->precision = p;
        a->wi = (int64_t *)malloc(sizeof(int64_t) * window * out_size);
        for (size_t i = 0; i < window * out_size; i++) {
            double v = a->w[i] * ldexp(1.0, p);
            a->wi[i] = (int64_t)(v < 0 ? -floor(-v + 0.5) : floor(v + 0.5));  /* f64::round: half away from zero */
        }
    }
}

static uint16_t axis_apply(const orc_axis *a, size_t o, const uint16_t *src, size_t stride, long base) {
    const long x0 = a->start[o], n = a->count[o];
    if (a->wi) {
        int64_t ss = a->precision > 0 ? ((int64_t)1 << (a->precision - 1)) : 0;
        for (long k = 0; k < n; k++) ss += (int64_t)src[(size_t)(x0 - base + k) * stride] * a->wi[o * a->window + (size_t)k];
        ss >>= a->precision;
        return (uint16_t)(ss < 0 ? 0 : ss > 65535 ? 65535 : ss);
    }
    double acc = 0.0;
    for (long k = 0; k < n; k++) acc += a->w[o * a->window + (size_t)k] * (double)src[(size_t)(x0 - base + k) * stride];
    double v = floor(acc + 0.5); if (v < 0) v = 0; if (v > 65535) v = 65535;
    return (uint16_t)v;
}

/* mode 0: f64 taps normalised before the sum (one division per tap instead of one per pixel: differs from
 * resize_lanczos3_u16 by f64 rounding only);  mode 1 / 2: fixed point, see above */
ORC_API void orc_resize_crop_u16(const uint16_t *img, size_t Hh, size_t W, double left, double top, double cw, double ch,
                                 size_t dw, size_t dh, int mode, uint16_t *dst) {
    const int bits = mode == 0 ? 0 : mode == 1 ? -1 : 16;
    orc_axis ax, ay;
    axis_build(W, left, cw, dw, bits, &ax);
    axis_build(Hh, top, ch, dh, bits, &ay);
    long y_lo = dh ? ay.start[0] : 0, y_hi = dh ? ay.start[dh - 1] + ay.count[dh - 1] : 0;
    for (size_t o = 0; o < dh; o++) { if (ay.start[o] < y_lo) y_lo = ay.start[o]; if (ay.start[o] + ay.count[o] > y_hi) y_hi = ay.start[o] + ay.count[o]; }
    size_t nrows = (size_t)(y_hi - y_lo);
    uint16_t *tmp = (uint16_t *)malloc(sizeof(uint16_t) * (nrows ? nrows : 1) * (dw ? dw : 1));
    for (size_t r = 0; r < nrows; r++)
        for (size_t ox = 0; ox < dw; ox++) tmp[r * dw + ox] = axis_apply(&ax, ox, img + ((size_t)y_lo + r) * W, 1, 0);
    for (size_t oy = 0; oy < dh; oy++)
        for (size_t ox = 0; ox < dw; ox++) dst[oy * dw + ox] = axis_apply(&ay, oy, tmp + ox, dw, y_lo);
    free(tmp); axis_free(&ax); axis_free(&ay);
}

/* The whole image as the crop box (what a pre-built mip level is): lod_h x lod_w, row 0 = lowest frequency.
 * mode -1: the very function the tile encoder uses (resize_lanczos3_u16); 0/1/2 as orc_resize_crop_u16. */
ORC_API void orc_resize_whole_image(const uint16_t *img, size_t Hh, size_t W, uint32_t level_x, uint32_t level_y,
                                    int mode, uint16_t *dst) {
    orc_tile_geom g;
    orc_spectrogram_tile_geom(W, Hh, level_x, level_y, 0, 0, &g);
    if (mode < 0) resize_lanczos3_u16(img, W, Hh, 0.0, 0.0, (double)W, (double)Hh, g.lod_w, g.lod_h, dst);
    else orc_resize_crop_u16(img, Hh, W, 0.0, 0.0, (double)W, (double)Hh, g.lod_w, g.lod_h, mode, dst);
}

/* The LOD pixels of one tile before the row flip and the colour map (render_tiles.rs:330-338), resampled with
 * the given mode (-1: resize_lanczos3_u16, as orc_encode_spectrogram_tile).  Returns width * height. */
ORC_API size_t orc_spectrogram_tile_u16(const uint16_t *img, size_t Hh, size_t W, uint32_t level_x, uint32_t level_y,
                                        uint32_t tile_x, uint32_t tile_y, int mode, uint16_t *dst, orc_tile_geom *geom) {
    orc_tile_geom g;
    orc_spectrogram_tile_geom(W, Hh, level_x, level_y, tile_x, tile_y, &g);
    if (geom) *geom = g;
    size_t w = g.width, h = g.height;
    if (w == 0 || h == 0) return 0;
    double left = (double)g.origin_x * (double)W / (double)g.lod_w;
    double top = (double)g.origin_y * (double)Hh / (double)g.lod_h;
    double right = (double)(g.origin_x + w) * (double)W / (double)g.lod_w;
    double bottom = (double)(g.origin_y + h) * (double)Hh / (double)g.lod_h;
    if (mode < 0) resize_lanczos3_u16(img, W, Hh, left, top, right - left, bottom - top, w, h, dst);
    else orc_resize_crop_u16(img, Hh, W, left, top, right - left, bottom - top, w, h, mode, dst);
    return w * h;
}

/* img: Hh x W u16 (row 0 = lowest frequency).  colormap: n_colors*4 RGBA bytes.
 * Returns bytes written (40 + w*h*4); out must hold 40 + 520*520*4. */
ORC_API size_t orc_encode_spectrogram_tile(const uint16_t *img, size_t Hh, size_t W,
                                           const uint8_t *colormap, size_t colormap_bytes,
                                           uint64_t revision, uint32_t level_x, uint32_t level_y,
                                           uint32_t tile_x, uint32_t tile_y, uint8_t *out) {
    orc_tile_geom g;
    orc_spectrogram_tile_geom(W, Hh, level_x, level_y, tile_x, tile_y, &g);
    put_u64(out, revision);
    put_u32(out + 8, g.width); put_u32(out + 12, g.height);
    put_u32(out + 16, level_x); put_u32(out + 20, level_y);
    put_u32(out + 24, tile_x); put_u32(out + 28, tile_y);
    put_u32(out + 32, g.origin_x); put_u32(out + 36, g.origin_y);
    size_t w = g.width, h = g.height;
    if (w == 0 || h == 0) return 40;
    uint16_t *lod = (uint16_t *)malloc(sizeof(uint16_t) * w * h);
    /* crop box in source coordinates  render_tiles.rs:382-386 */
    double left = (double)g.origin_x * (double)W / (double)g.lod_w;
    double top = (double)g.origin_y * (double)Hh / (double)g.lod_h;
    double right = (double)(g.origin_x + w) * (double)W / (double)g.lod_w;
    double bottom = (double)(g.origin_y + h) * (double)Hh / (double)g.lod_h;
    if (level_x == 0 && level_y == 0) { /* integral crop == dst size: exact copy (pinned :464-471) */
        for (size_t y = 0; y < h; y++)
            memcpy(lod + y * w, img + (g.origin_y + y) * W + g.origin_x, sizeof(uint16_t) * w);
    } else {
        resize_lanczos3_u16(img, W, Hh, left, top, right - left, bottom - top, w, h, lod);
    }
    size_t color_count = colormap_bytes / 4;
    uint8_t *p = out + 40;
    for (size_t r = 0; r < h; r++) {                      /* rows reversed :340 */
        const uint16_t *row = lod + (h - 1 - r) * w;
        for (size_t x = 0; x < w; x++) {
            size_t ci = color_count <= 1 ? 0 : ((size_t)row[x] * (color_count - 1) + 32767) / 65535; /* :342-346 */
            memcpy(p, colormap + ci * 4, 4); p += 4;
        }
    }
    free(lod);
    return 40 + w * h * 4;
}

/* ------------------------------------------------------------------ */
/* CPU-baseline helper (bench.py cpu_baseline leg): one track through   */
/* the same step the GPU bench times — calc_spec (f32 FFT) -> min/max   */
/* -> dB range -> u16 image -> every level-0 RGBA tile — in one call,   */
/* so the Python driver only fans tracks out over threads the way the   */
/* reference fans channels out over rayon (core/mod.rs:152-163).        */
/* Returns the number of frames processed.                              */
/* ------------------------------------------------------------------ */
ORC_API size_t orc_track_step(const float *x, size_t n, size_t win, size_t hop, size_t n_fft,
                              const uint8_t *colormap, size_t colormap_bytes, float dB_range,
                              uint64_t *checksum) {
    size_t F = n_fft / 2 + 1;
    size_t T = orc_stft_n_frames(n, win, hop);
    if (T == 0) return 0;
    float *window = (float *)malloc(sizeof(float) * win);
    orc_calc_normalized_win(win, n_fft, window);
    float *spec = (float *)malloc(sizeof(float) * T * F);
    size_t T2 = orc_calc_spec(x, n, win, hop, n_fft, window, NULL, 0, 1, spec, NULL);
    free(window);
    if (T2 != T) { free(spec); return 0; }
    float mn, mx, lo, hi;
    orc_find_min_max(spec, T * F, &mn, &mx);
    orc_global_db_range(&mn, &mx, 1, dB_range, &lo, &hi);
    uint16_t *img = (uint16_t *)malloc(sizeof(uint16_t) * T * F);
    orc_convert_spectrogram_to_img(spec, T, F, 0, F, lo, hi, (uint32_t)(colormap_bytes / 4), img);
    free(spec);
    uint8_t *tile = (uint8_t *)malloc(40 + 520 * 520 * 4);
    uint64_t sum = 0;
    for (uint32_t tx = 0;; tx++) {
        int any = 0;
        for (uint32_t ty = 0;; ty++) {
            size_t len = orc_encode_spectrogram_tile(img, F, T, colormap, colormap_bytes, 1, 0, 0, tx, ty, tile);
            if (len <= 40) break;
            any = 1;
            sum += tile[40] + tile[len - 1] + len;
        }
        if (!any) break;
    }
    free(tile); free(img);
    if (checksum) *checksum = sum;
    return T;
}
