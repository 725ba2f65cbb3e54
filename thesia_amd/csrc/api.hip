// api.hip — C ABI: errors, host helpers, device context, SpectrogramAnalyzer plan and the
// device-pointer ("layer A") entry points.  The TrackManager mirror lives in track_manager.hip.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <mutex>
#include <vector>

#include "common.h"
#include "context.h"
#include "host_math.h"
#include "kernels.h"
#include "mel_fuse.h"
#if !defined(TH_MEL_BAND_TAPS_2048)
#define TH_MEL_BAND_TAPS_2048 64u  // (measured thresholds: th_plan_create)
#define TH_MEL_BAND_TAPS_1024 80u
#endif
#if !defined(TH_MEL_BAND_SPREAD)
#define TH_MEL_BAND_SPREAD 1  // the banded tables' first bins on distinct LDS banks (mel_fuse.h; 0: as the filters start, for A/B runs)
#endif

// ------------------------------------------------------------------------------------------ errors
namespace th {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}
const char *get_error() { return g_err; }
}  // namespace th

using namespace th;

TH_API const char *th_last_error(void) { return th::get_error(); }
TH_API int th_version(void) { return 100; }

TH_API int th_device_count(int *count) {
    TH_TRY
    TH_REQUIRE(count, "count is NULL");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
    *count = n;
    return TH_OK;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ host helpers
TH_API int th_calc_framing_params(double win_ms, uint32_t t_overlap, uint32_t f_overlap, uint32_t sr, size_t *hop,
                                  size_t *win, size_t *n_fft) {
    TH_TRY
    TH_REQUIRE(hop && win && n_fft, "output pointer is NULL");
    // the reference asserts these in set_spec_setting (src-tauri/src/lib.rs:275-277)
    TH_REQUIRE(win_ms > 0. && t_overlap >= 1 && f_overlap >= 1, "win_ms > 0, t_overlap >= 1, f_overlap >= 1 required");
    calc_framing_params(win_ms, t_overlap, f_overlap, sr, hop, win, n_fft);
    return TH_OK;
    TH_CATCH
}

TH_API int th_stft_n_frames(size_t n_samples, size_t win, size_t hop, size_t *n_frames) {
    TH_TRY
    TH_REQUIRE(n_frames, "n_frames is NULL");
    TH_REQUIRE(win >= 1 && hop >= 1, "win and hop must be >= 1");
    *n_frames = stft_n_frames(n_samples, win, hop);
    return TH_OK;
    TH_CATCH
}

TH_API int th_calc_normalized_win(size_t win, size_t n_fft, float *out) {
    TH_TRY
    TH_REQUIRE(out, "out is NULL");
    TH_REQUIRE(win > 1 && n_fft >= 1, "win must be > 1");  // windows.rs:73 debug_assert!(size > 1)
    const std::vector<float> w = normalized_hann(win, n_fft);
    std::memcpy(out, w.data(), sizeof(float) * win);
    return TH_OK;
    TH_CATCH
}

TH_API int th_calc_mel_fb(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, int do_norm, float *out) {
    TH_TRY
    TH_REQUIRE(out, "out is NULL");
    TH_REQUIRE(n_fft % 2 == 0 && n_mel != 0, "n_fft must be even and n_mel non-zero");  // lib.rs:59-60
    const std::vector<float> fb = calc_mel_fb(sr, n_fft, n_mel, fmin, fmax, do_norm != 0);
    std::memcpy(out, fb.data(), sizeof(float) * fb.size());
    return TH_OK;
    TH_CATCH
}

TH_API int th_mel_default_n_mel(uint32_t sr, size_t n_fft, size_t *n_mel) {
    TH_TRY
    TH_REQUIRE(n_mel, "n_mel is NULL");
    TH_REQUIRE(sr > 0 && n_fft >= 2 && n_fft % 2 == 0, "sr > 0 and even n_fft >= 2 required");
    *n_mel = mel_default_n_mel(sr, n_fft);
    return TH_OK;
    TH_CATCH
}

TH_API int th_hz_range_to_idx(int freq_scale, float hz_min, float hz_max, uint32_t sr, size_t n, size_t *i_start,
                              size_t *i_end) {
    TH_TRY
    TH_REQUIRE(i_start && i_end, "output pointer is NULL");
    TH_REQUIRE(freq_scale == TH_FREQ_LINEAR || freq_scale == TH_FREQ_MEL, "bad freq_scale %d", freq_scale);
    hz_range_to_idx(freq_scale, hz_min, hz_max, sr, n, i_start, i_end);
    return TH_OK;
    TH_CATCH
}

TH_API int th_global_db_range(const float *mins, const float *maxs, size_t n, float dB_range, float *min_dB,
                              float *max_dB) {
    TH_TRY
    TH_REQUIRE(min_dB && max_dB && (n == 0 || (mins && maxs)), "NULL pointer");
    global_db_range(mins, maxs, n, dB_range, min_dB, max_dB);
    return TH_OK;
    TH_CATCH
}

TH_API int th_shard_assign(const uint64_t *weights, size_t n_units, uint32_t world, uint32_t *owner) {
    TH_TRY
    TH_REQUIRE(world >= 1, "world must be >= 1");
    TH_REQUIRE(n_units == 0 || (weights && owner), "NULL pointer");
    shard_assign(weights, n_units, world, owner);
    return TH_OK;
    TH_CATCH
}

TH_API int th_spectrogram_tile_geometry(size_t W, size_t Hh, uint32_t lx, uint32_t ly, uint32_t tx, uint32_t ty,
                                        th_tile_geom *geom) {
    TH_TRY
    TH_REQUIRE(geom, "geom is NULL");
    const TileGeom g = spectrogram_tile_geometry(W, Hh, lx, ly, tx, ty);
    geom->width = (uint32_t)g.width;
    geom->height = (uint32_t)g.height;
    geom->origin_x = (uint32_t)g.origin_x;
    geom->origin_y = (uint32_t)g.origin_y;
    geom->lod_width = g.lod_w;
    geom->lod_height = g.lod_h;
    return TH_OK;
    TH_CATCH
}

TH_API int th_waveform_tile_geometry(size_t n_samples, uint32_t level, uint32_t tile_index, size_t *start,
                                     size_t *bin_count, size_t *samples_per_bin) {
    TH_TRY
    TH_REQUIRE(start && bin_count && samples_per_bin, "output pointer is NULL");
    waveform_tile_geometry(n_samples, level, tile_index, start, bin_count, samples_per_bin);
    return TH_OK;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ context
TH_API int th_ctx_create(int device, void *hip_stream, th_ctx **out) {
    return th_ctx_create_ex(device, hip_stream, hip_stream != nullptr, out);
}

TH_API int th_ctx_create_ex(int device, void *hip_stream, int use_given_stream, th_ctx **out) {
    TH_TRY
    TH_REQUIRE(out, "out is NULL");
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0)
        return fail(TH_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
    TH_REQUIRE(device >= 0 && device < n, "device %d out of range (have %d)", device, n);
    TH_HIP(hipSetDevice(device));
    th_ctx *c = new th_ctx();
    c->device = device;
    {
        int n_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && n_cu > 0)
            c->n_cu = (uint32_t)n_cu;
    }
    if (use_given_stream) {
        c->stream = reinterpret_cast<hipStream_t>(hip_stream);  // NULL = legacy default stream
        c->own_stream = false;
    } else {
        hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete c;
            return fail(TH_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
        }
        c->own_stream = true;
    }
    if (hipEventCreate(&c->ev0) != hipSuccess || hipEventCreate(&c->ev1) != hipSuccess) {
        delete c;
        return fail(TH_ERR_HIP, "hipEventCreate failed");
    }
    *out = c;
    return TH_OK;
    TH_CATCH
}

TH_API int th_ctx_destroy(th_ctx *c) {
    TH_TRY
    if (!c) return TH_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    c->release_scratch();
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
    return TH_OK;
    TH_CATCH
}

TH_API int th_ctx_synchronize(th_ctx *c) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
    TH_CATCH
}

struct th_graph {
    th_ctx *ctx = nullptr;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

TH_API int th_ctx_capture_begin(th_ctx *c) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    return TH_OK;
    TH_CATCH
}

TH_API int th_ctx_capture_end(th_ctx *c, th_graph **out) {
    TH_TRY
    TH_REQUIRE(c && out, "NULL argument");
    *out = nullptr;
    TH_HIP(hipSetDevice(c->device));
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (e != hipSuccess || g == nullptr) {
        (void)hipGetLastError();
        if (g) (void)hipGraphDestroy(g);
        return fail(TH_ERR_HIP, "stream capture failed or was invalidated: %s", hipGetErrorString(e));
    }
    hipGraphExec_t x = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e2 != hipSuccess) {
        (void)hipGraphDestroy(g);
        return fail(TH_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e2));
    }
    th_graph *gr = new th_graph;
    gr->ctx = c;
    gr->graph = g;
    gr->exec = x;
    *out = gr;
    return TH_OK;
    TH_CATCH
}

TH_API int th_graph_launch(th_graph *g) {
    TH_TRY
    TH_REQUIRE(g && g->exec, "graph is NULL");
    TH_HIP(hipSetDevice(g->ctx->device));
    TH_HIP(hipGraphLaunch(g->exec, g->ctx->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_graph_destroy(th_graph *g) {
    TH_TRY
    if (!g) return TH_OK;
    if (g->exec) (void)hipGraphExecDestroy(g->exec);
    if (g->graph) (void)hipGraphDestroy(g->graph);
    delete g;
    return TH_OK;
    TH_CATCH
}

TH_API int th_dev_alloc(th_ctx *c, size_t bytes, void **dptr) {
    TH_TRY
    TH_REQUIRE(c && dptr, "NULL argument");
    *dptr = nullptr;
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return TH_OK;
    TH_CATCH
}
TH_API int th_dev_free(th_ctx *c, void *dptr) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (dptr) {
        TH_HIP(hipSetDevice(c->device));
        TH_HIP(hipStreamSynchronize(c->stream));
        TH_HIP(hipFree(dptr));
    }
    return TH_OK;
    TH_CATCH
}
TH_API int th_dev_upload(th_ctx *c, void *dst, const void *src, size_t bytes) {
    TH_TRY
    TH_REQUIRE(c && (bytes == 0 || (dst && src)), "NULL argument");
    if (bytes) {
        TH_HIP(hipSetDevice(c->device));
        TH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
        TH_HIP(hipStreamSynchronize(c->stream));
    }
    return TH_OK;
    TH_CATCH
}
TH_API int th_dev_download(th_ctx *c, void *dst, const void *src, size_t bytes) {
    TH_TRY
    TH_REQUIRE(c && (bytes == 0 || (dst && src)), "NULL argument");
    if (bytes) {
        TH_HIP(hipSetDevice(c->device));
        TH_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
        TH_HIP(hipStreamSynchronize(c->stream));
    }
    return TH_OK;
    TH_CATCH
}
TH_API int th_dev_copy(th_ctx *c, void *d_dst, const void *d_src, size_t bytes) {
    TH_TRY
    TH_REQUIRE(c && (bytes == 0 || (d_dst && d_src)), "NULL argument");
    TH_REQUIRE(((reinterpret_cast<uintptr_t>(d_dst) | reinterpret_cast<uintptr_t>(d_src) | bytes) & 15u) == 0,
               "pointers and size must be multiples of 16 bytes");
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(launch_copy_f4(d_src, d_dst, bytes, c->stream));
    return TH_OK;
    TH_CATCH
}
TH_API int th_timer_start(th_ctx *c) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(hipEventRecord(c->ev0, c->stream));
    return TH_OK;
    TH_CATCH
}
TH_API int th_timer_stop_ms(th_ctx *c, float *ms) {
    TH_TRY
    TH_REQUIRE(c && ms, "NULL argument");
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(hipEventRecord(c->ev1, c->stream));
    TH_HIP(hipEventSynchronize(c->ev1));
    TH_HIP(hipEventElapsedTime(ms, c->ev0, c->ev1));
    return TH_OK;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ plan
namespace th {

int DeviceTable::ensure(size_t bytes) {
    if (bytes <= cap) return TH_OK;
    if (dptr) {
        TH_HIP(hipFree(dptr));
        dptr = nullptr;
        cap = 0;
    }
    size_t want = bytes < 4096 ? 4096 : bytes + bytes / 2;
    TH_HIP(hipMalloc(&dptr, want));
    cap = want;
    last.clear();
    return TH_OK;
}

// Upload `bytes` of `src` unless the table already holds exactly these bytes (bench loops and
// repeated tile requests re-send identical descriptor tables).  Synchronous on purpose: the
// previous launch may still be reading the old table.
int DeviceTable::upload(hipStream_t s, const void *src, size_t bytes) {
    if (bytes == last.size() && bytes && std::memcmp(last.data(), src, bytes) == 0) return TH_OK;
    TH_HIP(hipStreamSynchronize(s));
    int rc = ensure(bytes);
    if (rc != TH_OK) return rc;
    if (bytes) TH_HIP(hipMemcpy(dptr, src, bytes, hipMemcpyHostToDevice));
    last.assign(static_cast<const unsigned char *>(src), static_cast<const unsigned char *>(src) + bytes);
    return TH_OK;
}

void DeviceTable::release() {
    if (dptr) (void)hipFree(dptr);
    dptr = nullptr;
    cap = 0;
    last.clear();
}

}  // namespace th

bool th_plan::use_wave() const {
    if (kernel_choice == 1) return false;
    return th::stft_wave_supported(g) && (g.n_mel == 0 || d_mel_bt != nullptr);
}
// mel plans on the wave kernel: the filterbank fused into the FFT kernel's epilogue where its table fits (n_fft 1024 /
// 2048 at the default launch shape), else amplitude out of the FFT kernel and the filterbank on the matrix cores
int th_plan::long_plan() const {
    if (kernel_choice == 14) return 1;
    if (kernel_choice == 15) return 2;
    // Mel plans at n_fft 16384 that have a moment table: the workgroup-per-frame kernel with the filterbank in its epilogue at EVERY
    // hop.  For linear rows the subwave plan is the faster one at hops other than n_fft / 4 (0.63 against 0.67 ms at 16320 / 4080,
    // the UI's 340 ms window), but it has no epilogue and would need the second kernel: 0.99 against 0.82 ms there, 1.23 against 1.04
    // at 12000 / 3000, 1.79 against 1.50 at 16320 / 2040 (64 ch x 30 s).  Selector 12 (the two kernels) keeps the size's default.
    if (g.n_mel != 0 && g.log2_nc == 13 && d_mel_mom != nullptr && kernel_choice != 12) return 1;
    return 0;
}
bool th_plan::use_mel_fused() const {
    if (g.n_mel == 0 || !use_wave() || kernel_choice == 3 || kernel_choice == 7) return false;
    // n_fft 4096 (round 6): the moment form of the filterbank as the FFT kernel's epilogue — no table in LDS (the eight slabs
    // leave none), any mel count; instantiated for hop 1024 and for the even-offset grid-aligned shapes of the 96 / 88.2 kHz
    // defaults (not when selector 4 switches that mode off).  Selector 12 keeps round 5's two kernels (A/B).
    if (g.log2_nc == 11) return kernel_choice != 12 && d_mel_mom != nullptr && th::stft_wave_mel_fits(g, wave_waves, 1, true);
    // n_fft 8192 / 16384 (round 6): the same moment form as the epilogue of the workgroup-per-frame kernel, where that kernel is the
    // plan that runs (long_plan(): at 16384 a mel plan with a table takes the block kernel at every hop; not under selector 15); selector 12 keeps the two kernels
    if (g.log2_nc == 12 || g.log2_nc == 13)
        return kernel_choice != 12 && d_mel_mom != nullptr && th::stft_block_mel_fused_applies(g, long_plan());
    // n_fft 512 under narrow filters: the banded sums of mel_rows_kernel as the epilogue of the four-frames-per-wave kernel
    if (g.log2_nc == 8) return (d_mel_rows != nullptr && th::stft_wave_multi_mel_fits(g, wave_waves, mel_rows_groups * (uint32_t)(th::MEL_ROWS_W + 1) * 64u)) || use_mel_moment_small();
    return mel_bsum_fits() || (d_mel_fuse != nullptr && th::stft_wave_mel_fits(g, wave_waves, mel_fuse_words)) || use_mel_moment_small();
}
// n_fft 1024 / 2048 under more mels than an LDS table form holds (512: low sample rates under long windows, or f_overlap 2 / 4): the moment form
// with its table in global memory, as at n_fft 4096 (one-frame epilogue; any frame loop of the size).  Selectors 12 / 2 keep the two kernels.
bool th_plan::use_mel_moment_small() const {
    if (g.n_mel == 0 || g.log2_nc < 8 || g.log2_nc > 10 || d_mel_mom == nullptr || kernel_choice == 12 || kernel_choice == 8 || kernel_choice == 3 || kernel_choice == 7) return false;
    // n_fft 512 (four frames per wave): where the banded rows table does not exist — filters wider than its 8 bins: 10 / 20 ms windows at 16 .. 48 kHz
    if (g.log2_nc == 8) return d_mel_rows == nullptr && th::stft_wave_multi_applies(g, 2) && th::stft_wave_multi_mel_fits(g, wave_waves, 0);
    if (mel_bsum_fits() || (d_mel_fuse != nullptr && th::stft_wave_mel_fits(g, wave_waves, mel_fuse_words))) return false;
    return th::stft_wave_mel_fits(g, wave_waves, 0, true);
}
bool th_plan::mel_bsum_fits() const {
    return kernel_choice != 8 && d_mel_bsum != nullptr && th::stft_wave_mel_fits(g, wave_waves, mel_bsum_words, true);
}
bool th_plan::use_mel_mfma() const { return g.n_mel != 0 && use_wave() && !use_mel_fused(); }

static void plan_free(th_plan *p) {
    if (!p) return;
    for (hipEvent_t e : p->ev_k0) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->ev_k1) (void)hipEventDestroy(e);
    if (p->d_wtab) (void)hipFree(p->d_wtab);
    if (p->d_wtab_phased) (void)hipFree(p->d_wtab_phased);
    if (p->d_queue_head) (void)hipFree(p->d_queue_head);
    if (p->d_mel_bt) (void)hipFree(p->d_mel_bt);
    if (p->d_mel_band) (void)hipFree(p->d_mel_band);
    if (p->d_mel_slice) (void)hipFree(p->d_mel_slice);
    if (p->d_mel_rows) (void)hipFree(p->d_mel_rows);
    if (p->d_mel_fuse) (void)hipFree(p->d_mel_fuse);
    if (p->d_mel_bsum) (void)hipFree(p->d_mel_bsum);
    if (p->d_mel_mom) (void)hipFree(p->d_mel_mom);
    p->amp_buf.release();
    p->chunk_mm.release();
    p->post_jobs.release();
    p->mel_jobs.release();
    p->mel_tile_start.release();
    if (p->d_window) (void)hipFree(p->d_window);
    if (p->d_bs_chirp) (void)hipFree(p->d_bs_chirp);
    if (p->d_bs_bhat) (void)hipFree(p->d_bs_bhat);
    if (p->d_bs_twm) (void)hipFree(p->d_bs_twm);
    if (p->d_bs_tws) (void)hipFree(p->d_bs_tws);
    if (p->d_tw) (void)hipFree(p->d_tw);
    if (p->d_twc) (void)hipFree(p->d_twc);
    if (p->d_mel_fb) (void)hipFree(p->d_mel_fb);
    if (p->d_mel_lo) (void)hipFree(p->d_mel_lo);
    if (p->d_mel_hi) (void)hipFree(p->d_mel_hi);
    p->jobs.release();
    p->gen_scratch.release();
    p->tile_start.release();
    p->edge_jobs.release();
    p->edge_tile_start.release();
    delete p;
}

TH_API int th_plan_create(th_ctx *c, uint32_t sr, size_t win, size_t hop, size_t n_fft, int freq_scale,
                          size_t n_mel, th_plan **out) {
    TH_TRY
    TH_REQUIRE(c && out, "NULL argument");
    *out = nullptr;
    TH_REQUIRE(freq_scale == TH_FREQ_LINEAR || freq_scale == TH_FREQ_MEL, "bad freq_scale %d", freq_scale);
    TH_REQUIRE(win > 1 && hop >= 1 && win <= n_fft, "need 1 < win <= n_fft and hop >= 1 (win=%zu hop=%zu n_fft=%zu)",
               win, hop, n_fft);
    TH_REQUIRE(sr > 0, "sr must be > 0");
    // Any power of two from 2 to 2^20: what SpecSetting::calc_framing_params (spectrogram.rs:56-98) yields for every window
    // the UI accepts with f_overlap a power of two — winMillisec has a lower bound only (tracks.ts:205, Control.tsx:96-107),
    // so e.g. 400 ms at 48 kHz is n_fft 32768 and 1 ms at 4 kHz is n_fft 4.  (The reference's realfft takes any length: a
    // non-power-of-two n_fft needs f_overlap = 3, 5, ..., which no control offers.)
    // Round 5: n_fft = 2^a * odd with a >= 1 and odd <= 63 — f_overlap = 3, 5, 6, 7, ... (spectrogram.rs:66-72: n_fft =
    // next_pow2(win) * f_overlap; no UI control offers them, the reference's realfft plans any length) run on the generic
    // kernel, which takes the odd factor as one more Stockham pass.
    size_t odd_part = n_fft;
    while (odd_part > 1 && odd_part % 2 == 0) odd_part /= 2;
    // Round 6: an odd factor above 63 (f_overlap = 67, 71, 130, ...) runs as a chirp-z convolution (stft_bluestein_kernel): every
    // even n_fft up to TH_MAX_N_FFT is planned.  (n_fft = next_pow2(win) * f_overlap is even for every win > 1.)
    if (n_fft < 2 || n_fft % 2 != 0 || n_fft > TH_MAX_N_FFT)
        return fail(TH_ERR_UNSUPPORTED, "n_fft=%zu: supported is every even n_fft from 2 to %u", n_fft, (unsigned)TH_MAX_N_FFT);
    TH_HIP(hipSetDevice(c->device));

    th_plan *p = new th_plan();
    p->ctx = c;
    p->sr = sr;
    p->freq_scale = freq_scale;
    StftGeom &g = p->g;
    g.hop = (uint32_t)hop;
    g.win = (uint32_t)win;
    g.n_fft = (uint32_t)n_fft;
    g.pad_left = (uint32_t)((n_fft - win) / 2);
    g.nc = (uint32_t)(n_fft / 2);
    g.odd_m1 = (uint32_t)odd_part - 1u;
    g.log2_nc = ilog2(g.nc / (uint32_t)odd_part);  // log2 of the power-of-two part of nc
    g.n_freq = (uint32_t)(n_fft / 2 + 1);
    g.n_mel = 0;
    g.height = g.n_freq;
    g.frames_per_tile = 8;  // generic kernel; the wave kernel's tile is set at launch

    // window (windows.rs) and twiddles W_{n_fft}^i (double → f32)
    const std::vector<float> w = normalized_hann(win, n_fft);
    std::vector<cf32> tw(n_fft);
    for (size_t i = 0; i < n_fft; i++) {
        const double a = -2.0 * M_PI * (double)i / (double)n_fft;
        tw[i].re = (float)std::cos(a);
        tw[i].im = (float)std::sin(a);
    }
    int rc = TH_OK;
    auto up = [&](void **d, const void *h, size_t bytes) -> int {
        TH_HIP(hipMalloc(d, bytes));
        TH_HIP(hipMemcpy(*d, h, bytes, hipMemcpyHostToDevice));
        return TH_OK;
    };
    {
        // wave-kernel window table: wtab[n] = 0.5 * (wpad[2n], wpad[2n+1]), wpad = zero-padded window
        // (the 1/2 of the real-FFT split pass and the kernel's 2^32 pre-scale (stft_wave.h: WAVE_PRESCALE, undone in the
        // dB conversion) are folded in here; exact, powers of two)
        const float half = 0.5f * th::WAVE_PRESCALE;
        std::vector<float> wpad(n_fft, 0.f);
        for (size_t i = 0; i < win; i++) wpad[g.pad_left + i] = half * w[i];
        rc = up((void **)&p->d_wtab, wpad.data(), wpad.size() * sizeof(float));
        // grid-aligned modes of the wave kernel (kernels_stft.hip): the window at offset 0 instead of pad_left.
        // phased (hop 480): behind 48 zero pairs (read 0, 96, 64 or 32 samples lower);  dynamic (e.g. hop 441): behind 64
        // zero pairs, followed by the same with every pair shifted by one sample (odd offsets)
        const int pm = th::stft_wave_phased_mode(g, 0);
        if (rc == TH_OK && pm == 1) {
            std::vector<float> w4(96 + n_fft, 0.f);
            for (size_t i = 0; i < win; i++) w4[96 + i] = half * w[i];
            rc = up((void **)&p->d_wtab_phased, w4.data(), w4.size() * sizeof(float));
        } else if (rc == TH_OK && (pm == 2 || pm == 3)) {  // (3 reads the even table only: the same buffer)
            std::vector<float> t0(n_fft + 2, 0.f), w2(2 * (128 + n_fft), 0.f);
            for (size_t i = 0; i < win; i++) t0[i] = half * w[i];
            for (size_t i = 0; i < n_fft; i++) {
                w2[128 + i] = t0[i];                       // even table: pairs (t0[2n], t0[2n+1])
                w2[128 + n_fft + 128 + i] = t0[i + 1];     // odd table:  pairs (t0[2n+1], t0[2n+2])
            }
            rc = up((void **)&p->d_wtab_phased, w2.data(), w2.size() * sizeof(float));
        }
    }
    if (rc == TH_OK) rc = up((void **)&p->d_window, w.data(), w.size() * sizeof(float));
    if (rc == TH_OK) {
        const uint32_t zero[4] = {0, 0, 0, 0};  // [0] chunk queue head (rewound by wave_post_kernel after every launch), [1..3] spare
        rc = up((void **)&p->d_queue_head, zero, sizeof zero);
    }
    if (rc == TH_OK) rc = up((void **)&p->d_tw, tw.data(), tw.size() * sizeof(cf32));
    if (rc == TH_OK && odd_part > th::STFT_MAX_DIRECT_ODD) {  // Bluestein plan: chirp, FFT_M(b), twiddles of the M-point passes and of the split pass (doubles)
        const BluesteinTables bt = bluestein_tables(n_fft, th::stft_bluestein_m(g));
        rc = up((void **)&p->d_bs_chirp, bt.chirp.data(), bt.chirp.size() * sizeof(double));
        if (rc == TH_OK) rc = up((void **)&p->d_bs_bhat, bt.bhat.data(), bt.bhat.size() * sizeof(double));
        if (rc == TH_OK) rc = up((void **)&p->d_bs_twm, bt.twm.data(), bt.twm.size() * sizeof(double));
        if (rc == TH_OK) rc = up((void **)&p->d_bs_tws, bt.tws.data(), bt.tws.size() * sizeof(double));
    }
    if (rc == TH_OK && th::stft_subwave_applies(g)) {  // n_fft 32768: per-thread constants of stft_subwave_kernel's combining pass
        std::vector<cf32> twc(th::stft_subwave_twc_len(g));
        th::stft_subwave_build_twc(g, tw.data(), twc.data());
        rc = up((void **)&p->d_twc, twc.data(), twc.size() * sizeof(cf32));
    }
    if (rc == TH_OK && freq_scale == TH_FREQ_MEL) {
        if (n_mel == 0) n_mel = mel_default_n_mel(sr, n_fft);  // calc_mel_fb_default, lib.rs:91-103
        if (n_mel == 0 || n_mel > 65535) {
            plan_free(p);
            return fail(TH_ERR_INVALID_ARG, "n_mel=%zu out of range", n_mel);
        }
        // the dense (F x n_mel) filterbank of calc_mel_fb (src-common/src/lib.rs:46-89) is what the generic kernel reads: at the
        // default mel counts of very long windows it no longer fits anything (n_fft 131072 at 48 kHz: 65537 x ~22000 floats)
        if ((double)(n_fft / 2 + 1) * (double)n_mel * sizeof(float) > (double)((size_t)1 << 30)) {
            plan_free(p);
            return fail(TH_ERR_UNSUPPORTED, "mel filterbank of %zu bins x %zu mels exceeds 1 GiB", n_fft / 2 + 1, n_mel);
        }
        g.n_mel = (uint32_t)n_mel;
        g.height = (uint32_t)n_mel;
        p->h_mel_fb = calc_mel_fb(sr, n_fft, n_mel, 0.f, -1.f, true);
        std::vector<uint32_t> lo(n_mel, 0), hi(n_mel, 0);
        for (size_t m = 0; m < n_mel; m++) {
            uint32_t a = g.n_freq, b = 0;
            for (uint32_t f = 0; f < g.n_freq; f++)
                if (p->h_mel_fb[(size_t)f * n_mel + m] != 0.f) {
                    a = std::min(a, f);
                    b = f + 1;
                }
            lo[m] = b ? a : 0;
            hi[m] = b;
        }
        if (th::stft_wave_supported(g)) {   // (no wave / block kernel for this geometry: the generic kernel reads none of these)
            // MFMA mel path tables (kernels_mel.hip; any mel count since round 4 — the Mel default of long windows, e.g. 5571 mels
            // at n_fft 32768 / 48 kHz, used to fall back to the generic kernel: 12 ms where the linear plan takes 2): for every N tile j of 16 mels the band of K blocks (16 bins each)
            // that hold non-zeros, and the filterbank of those blocks in operand order: lane (kq = lane / 16,
            // li = lane % 16), step s -> fb[16 kb + 4 kq + s][16 j + li]
            const uint32_t nt = (uint32_t)((n_mel + 15) / 16);
            const uint32_t kb_n = (g.n_freq + 15) / 16;
            std::vector<uint32_t> band(3 * (size_t)nt, 0);
            std::vector<float> bt;
            auto fbv = [&](uint32_t f, size_t m) { return (f < g.n_freq && m < n_mel) ? p->h_mel_fb[(size_t)f * n_mel + m] : 0.f; };
            for (uint32_t j = 0; j < nt; j++) {
                // the tile's K band from the per-mel non-zero ranges [lo, hi) found above (the dense scan over
                // every (bin, mel) of every tile was ~90 M filterbank reads at the 5571-mel default of n_fft 32768)
                uint32_t klo = kb_n, khi = 0;
                for (size_t m = 16 * (size_t)j; m < std::min<size_t>(16 * (size_t)j + 16, n_mel); m++)
                    if (hi[m]) {
                        klo = std::min(klo, lo[m] / 16);
                        khi = std::max(khi, (hi[m] + 15) / 16);
                    }
                if (!khi) klo = 0;
                band[3 * j] = klo;
                band[3 * j + 1] = khi;
                band[3 * j + 2] = (uint32_t)(bt.size() / 256);
                for (uint32_t kb = klo; kb < khi; kb++)
                    for (uint32_t lane = 0; lane < 64; lane++)
                        for (uint32_t st = 0; st < 4; st++) bt.push_back(fbv(16 * kb + 4 * (lane >> 4) + st, 16 * (size_t)j + (lane & 15u)));
            }
            p->mel_zero_block = (uint32_t)(bt.size() / 256);
            bt.insert(bt.end(), 256, 0.f);
            p->mel_kblocks = kb_n;
            p->mel_ntiles = nt;
            // slices of the tile range (grid y): contiguous, about equal numbers of 4-K-block groups
#if !defined(TH_MEL_SLICES)
#define TH_MEL_SLICES 6
#endif
            std::vector<uint32_t> slice{0};
            {
                uint64_t total = 0;
                for (uint32_t j = 0; j < nt; j++) total += (band[3 * j + 1] - band[3 * j] + 3) / 4 + 1;  // + epilogue
                const uint32_t want = std::min<uint32_t>(TH_MEL_SLICES, nt);
                uint64_t acc_g = 0;
                for (uint32_t j = 0; j < nt; j++) {
                    acc_g += (band[3 * j + 1] - band[3 * j] + 3) / 4 + 1;
                    if (slice.size() < want && acc_g * want >= total * slice.size() && j + 1 < nt) slice.push_back(j + 1);
                }
                slice.push_back(nt);
            }
            p->mel_slices = (uint32_t)slice.size() - 1;
            {   // short rows under narrow filters (n_fft 512 at the default mel counts): the per-mel table of mel_rows_kernel
                uint32_t widest = 0;
                for (size_t m = 0; m < n_mel; m++) widest = std::max(widest, hi[m] - lo[m]);
                const uint32_t ng = (uint32_t)((n_mel + 63) / 64);
                // (only n_fft 512 plans on the wave kernel ever read it — ADVICE r3: n_fft 128 / 256 used to pay the upload too)
                if (g.log2_nc == 8 && th::stft_wave_supported(g) && kb_n <= (uint32_t)th::MEL_ROWS_NKB && g.n_freq - 1 + th::MEL_ROWS_W <= 16 * th::MEL_ROWS_NKB + 4 &&  // (reads stay in the LDS row)
                    widest <= (uint32_t)th::MEL_ROWS_W && ng <= (uint32_t)th::MEL_ROWS_MAX_GROUPS) {
                    constexpr uint32_t W = th::MEL_ROWS_W;
                    std::vector<uint32_t> tab((size_t)ng * (W + 1) * 64, 0u);
                    for (size_t m = 0; m < n_mel; m++) {
                        const size_t at = (m / 64) * (W + 1) * 64 + m % 64;
                        tab[at] = lo[m];
                        for (uint32_t t = 0; t < W && lo[m] + t < hi[m]; t++)
                            std::memcpy(&tab[at + (1 + t) * 64], &p->h_mel_fb[(size_t)(lo[m] + t) * n_mel + m], 4);
                    }
                    p->mel_rows_groups = ng;
                    rc = up((void **)&p->d_mel_rows, tab.data(), tab.size() * sizeof(uint32_t));
                }
            }
            if (rc == TH_OK) rc = up((void **)&p->d_mel_bt, bt.data(), bt.size() * sizeof(float));
            if (rc == TH_OK) rc = up((void **)&p->d_mel_band, band.data(), band.size() * sizeof(uint32_t));
            if (rc == TH_OK) rc = up((void **)&p->d_mel_slice, slice.data(), slice.size() * sizeof(uint32_t));
        }
        if (rc == TH_OK && th::stft_wave_supported(g) && n_mel <= 512) {  // (the fused forms: lane = mel in at most 8 groups of 64)
            // fused mel epilogue of the wave kernel: piece / gather tables, when the filterbank has the expected structure
            const th::MelFuseHost mf = th::build_mel_fuse(p->h_mel_fb.data(), g.n_freq, (uint32_t)n_mel, th::stft_wave_mel_max_pieces(g));
            if (rc == TH_OK && g.log2_nc == 11) {
                // n_fft 4096 (two kernels): the same banded table for mel_band_rows_kernel, where it fits LDS beside four rows
                // (the paired layout as in the wave kernels; with dword reads the bank-spread table had measured 3 % SLOWER here, 2.08 ->
                // 2.15 ms at the 96 kHz default, profiles/r04_ab_mel_bank_spread.txt — the paired one is 1 % faster than the plain table)
                const th::MelBandHost mb = th::build_mel_band(p->h_mel_fb.data(), g.n_freq, (uint32_t)n_mel, 1u << 16, false, TH_MEL_ROWS_PAIRED != 0);
                if (mb.ok && th::mel_band_rows_fits(g.n_freq, (uint32_t)mb.words.size())) {
                    p->mel_bsum_words = (uint32_t)mb.words.size();
                    std::copy(mb.words.begin(), mb.words.begin() + 16, p->mel_bsum_hdr);
                    p->mel_bsum_groups = mb.n_groups;
                    rc = up((void **)&p->d_mel_bsum, mb.words.data(), mb.words.size() * sizeof(uint32_t));
                }
            }
            if (rc == TH_OK && (g.log2_nc == 9 || g.log2_nc == 10)) {
                // (the wave kernels of n_fft 1024 / 2048 read the table's paired layout, mel_banded<true>)
                const th::MelBandHost mb = th::build_mel_band(p->h_mel_fb.data(), g.n_freq, (uint32_t)n_mel, 1u << 16, TH_MEL_BAND_SPREAD != 0, TH_MEL_BAND_PAIRED != 0);
                // Measured against the pieces / gather form (same box, alternating; taps = the sum over the groups of their
                // widest filter): n_fft 2048 — 56 / 60 / 64 taps (256 mels, the 44.1 / 48 kHz defaults) 6 / 8 / 10 % faster; 72 taps
                // (128 mels) 4 % faster in the register-reuse kernel (hop a multiple of 128: config 4), 6 % slower in the full-reload
                // one (hop 480); 96 (200 mels) 5 % slower, 124 (64 mels) 40 % slower; n_fft 1024 — 32 / 36 / 44 taps (22.05 kHz
                // default, 128 mels, 16 kHz default) 22 / 13 / 12 % faster, 76 (48 kHz, 80 mels) 1 % faster
                // (round 4: the table's first bins are spread over the LDS banks, which may add 4 taps to a group; the limits
                // were measured on, and still apply to, the tap count before that)
                const uint32_t taps = mb.taps_unshifted;
                const uint32_t max_taps = g.log2_nc == 9 ? TH_MEL_BAND_TAPS_1024 : (g.hop % 128 == 0 ? TH_MEL_BAND_TAPS_2048 + 8u : TH_MEL_BAND_TAPS_2048);
                if (mb.ok && taps <= max_taps) {
                    p->mel_bsum_words = (uint32_t)mb.words.size();
                    std::copy(mb.words.begin(), mb.words.begin() + 16, p->mel_bsum_hdr);
                    p->mel_bsum_groups = mb.n_groups;
                    p->mel_bsum_reach = mb.reach;
                    rc = up((void **)&p->d_mel_bsum, mb.words.data(), mb.words.size() * sizeof(uint32_t));
                }
            }
            if (mf.ok && rc == TH_OK) {
                p->mel_fuse_words = (uint32_t)mf.words.size();
                p->mel_fuse_slots = mf.n_slots;
                p->mel_fuse_groups = mf.n_groups;
                rc = up((void **)&p->d_mel_fuse, mf.words.data(), mf.words.size() * sizeof(uint32_t));
            }
        }
        // (n_fft 1024 / 2048: only where no LDS table form exists — more than 512 mels)
        const bool small_mom = ((g.log2_nc == 9 || g.log2_nc == 10) && p->d_mel_bsum == nullptr && p->d_mel_fuse == nullptr) || (g.log2_nc == 8 && p->d_mel_rows == nullptr);
        if (rc == TH_OK && (g.log2_nc == 11 || g.log2_nc == 12 || g.log2_nc == 13 || small_mom) && th::stft_wave_supported(g)) {
            // n_fft 4096 (the wave kernel) and 8192 / 16384 (the workgroup-per-frame kernel): the moment form (lane = segment of the
            // triangle points; mel_fuse.h) for the fused epilogue, any mel count
            std::vector<float> lin, mfp;
            mel_fb_points(sr, n_fft, n_mel, 0.f, -1.f, lin, mfp);
            const th::MelMomHost mm = th::build_mel_moments(p->h_mel_fb.data(), lin.data(), mfp.data(), g.n_freq, (uint32_t)n_mel,
                                                            g.log2_nc == 8 ? th::stft_wave_multi_amp_pitch() : g.log2_nc <= 11 ? 2u * (g.nc + g.nc / 16u) : th::stft_block_mel_max_index(g), g.log2_nc <= 11 && TH_MEL_BAND_SPREAD != 0);
            // (the workgroup-per-frame kernels walk a batch of groups in lockstep: a lane reads up to the table's widest group's
            // taps behind its first bin — inside the exchange buffer)
            const bool reach_ok = g.log2_nc <= 11 || (uint64_t)g.n_freq + mm.max_taps <= th::stft_block_mel_max_index(g);
            // (the workgroup-per-frame kernels: the same numbers at fixed addresses, build_mel_mom_lanes)
            const std::vector<uint32_t> lanes = g.log2_nc <= 11 ? std::vector<uint32_t>() : th::build_mel_mom_lanes(mm);
            const std::vector<uint32_t> &tab = g.log2_nc <= 11 ? mm.words : lanes;
            if (mm.ok && reach_ok && !tab.empty()) {
                p->mel_mom_groups = mm.n_groups;
                p->mel_mom_taps = mm.taps;
                p->mel_mom_max_dev = mm.max_dev;
                p->mel_mom_max_amp = mm.max_amp;
                rc = up((void **)&p->d_mel_mom, tab.data(), tab.size() * sizeof(uint32_t));
            }
        }
        if (rc == TH_OK) rc = up((void **)&p->d_mel_fb, p->h_mel_fb.data(), p->h_mel_fb.size() * sizeof(float));
        if (rc == TH_OK) rc = up((void **)&p->d_mel_lo, lo.data(), lo.size() * sizeof(uint32_t));
        if (rc == TH_OK) rc = up((void **)&p->d_mel_hi, hi.data(), hi.size() * sizeof(uint32_t));
    }
    if (rc != TH_OK) {
        plan_free(p);
        return rc;
    }
    *out = p;
    return TH_OK;
    TH_CATCH
}

TH_API int th_plan_destroy(th_plan *p) {
    TH_TRY
    if (!p) return TH_OK;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    plan_free(p);
    return TH_OK;
    TH_CATCH
}

TH_API int th_plan_dims(const th_plan *p, size_t *n_freq, size_t *height) {
    TH_TRY
    TH_REQUIRE(p, "plan is NULL");
    if (n_freq) *n_freq = p->g.n_freq;
    if (height) *height = p->g.height;
    return TH_OK;
    TH_CATCH
}

TH_API int th_plan_set_kernel(th_plan *p, int which) {
    TH_TRY
    TH_REQUIRE(p, "plan is NULL");
    // bits 0-7: 0 auto, 1 generic, 2 wave, 3 wave + matrix-core mel, 4 wave without the phased mode, 5 phased mode also with the
    // fused mel epilogue, 6 wave with the two-frames-per-wave plan at n_fft 1024, 7 as 3 with the matrix-core kernel also where
    // the banded-sum kernel for short rows under narrow filters is the default (n_fft 512), 8 fused mel epilogue in its pieces / gather
    // form where the banded sums are the default, 9 wave kernel with the packed-f32 pipeline (stft_pk.h) on the launch shape it is
    // instantiated for (n_fft 2048, hop = n_fft / 4, linear dB, default waves; elsewhere as 2), 11 wave kernel with the sweep chunk
    // schedule (4-frame chunks dealt out in order) on large batches of that same shape (A/B; elsewhere as 2; 10 is reserved: as 2),
    // 12 mel plans at n_fft 4096: the two kernels (FFT -> amplitude rows -> banded sums / matrix cores) where the moment-form
    // epilogue is the default (hop 1024, the 96 / 88.2 kHz defaults: round 5's route, A/B; elsewhere as 2);
    // 13 fused mel epilogue one frame at a time where the frame-pair form is the default (n_fft 2048 banded sums; A/B; elsewhere as 2);
    // 14 the workgroup-per-frame Stockham kernel (stft_block_kernel) where stft_subwave_kernel is the default (A/B; elsewhere as 2);
    // 15 stft_subwave_kernel wherever it exists (n_fft 8192 .. 32768) also where the block kernel is the default (A/B; elsewhere as 2);
    // bits 8-15 (tuning): waves per workgroup
    const int k = which & 0xff, wv = (which >> 8) & 0xff;
    TH_REQUIRE(k >= 0 && k <= 15, "kernel selector must be 0 .. 15");
    TH_REQUIRE(wv == 0 || wv == 4 || wv == 6 || wv == 7 || wv == 8 || wv == 10 || wv == 12 || wv == 14 || wv == 16,
               "waves per workgroup must be 4, 6, 7, 8, 10, 12, 14 or 16");
    // the multi-frame plans (n_fft 512; n_fft 1024 under selector 6) are instantiated for 8, 12 and 16 waves only
    const bool multi = p->g.n_mel == 0 ? (p->g.log2_nc == 8 || (p->g.log2_nc == 9 && k == 6)) : p->g.log2_nc == 8;
    if (multi && !(wv == 0 || wv == 8 || wv == 12 || wv == 16))
        return fail(TH_ERR_UNSUPPORTED, "the multi-frame wave kernel of n_fft %u runs with 8, 12 or 16 waves per workgroup, not %d",
                    p->g.n_fft, wv);
    // Variants that were measured and dropped are not in the product library (kernels.h, TH_AB_VARIANTS): refuse them by name
    // instead of a launch failure later
    if (!TH_AB_VARIANTS) {
        const char *what = nullptr;
        if (k == 9) what = "selector 9 (packed-f32 pipeline)";
        else if (k == 11) what = "selector 11 (sweep chunk schedule)";
        else if (k == 14 && p->g.log2_nc >= 14) what = "selector 14 at n_fft 32768 / 65536 (workgroup-per-frame Stockham kernels)";
        else if (k == 15 && p->g.log2_nc == 12) what = "selector 15 at n_fft 8192 (stft_subwave_kernel with four waves)";
        else if (!multi && p->g.log2_nc >= 9 && p->g.log2_nc <= 11 && wv != 0 && !(p->g.log2_nc == 11 ? (wv == 7 || wv == 8) : wv == 12))
            what = "a waves-per-workgroup shape other than the size's own (n_fft 1024 / 2048: 12; n_fft 4096: 8 or 7)";
        if (what) return fail(TH_ERR_UNSUPPORTED, "%s is an A/B variant: build with -DTH_AB_VARIANTS=1 (scripts/build_variant.sh)", what);
    }
    p->kernel_choice = k;
    p->wave_waves = wv;
    p->wave_chunk = (which >> 16) & 0xff;  // tuning: frames per chunk of the wave kernel (0 = default)
    return TH_OK;
    TH_CATCH
}

TH_API int th_build_ab_variants(void) { return TH_AB_VARIANTS ? 1 : 0; }

TH_API int th_plan_mel_moments_info(const th_plan *p, uint32_t *n_groups, uint32_t *taps, double *max_dev, double *max_amp) {
    TH_TRY
    TH_REQUIRE(p, "plan is NULL");
    const bool have = p->d_mel_mom != nullptr;
    if (n_groups) *n_groups = have ? p->mel_mom_groups : 0u;
    if (taps) *taps = have ? p->mel_mom_taps : 0u;
    if (max_dev) *max_dev = have ? p->mel_mom_max_dev : 0.0;
    if (max_amp) *max_amp = have ? p->mel_mom_max_amp : 0.0;
    return TH_OK;
    TH_CATCH
}

TH_API const char *th_plan_kernel_name(const th_plan *p) {
    if (!p) return "";
    if (p->use_mel_fused()) return th::stft_is_block_plan(p->g) ? "stft_block_kernel(fused mel)" : "stft_wave_kernel(fused mel)";
    if (p->use_mel_mfma() && p->d_mel_rows != nullptr && p->kernel_choice != 7) return "stft_wave_kernel+mel_rows_kernel";
    // (n_fft 32768, round 5: sixteen wave transforms + a combining pass, kernels_stft_long.hip; selector 14 keeps the block kernel)
    const bool subwave = th::stft_subwave_applies(p->g) && p->d_twc != nullptr && (p->long_plan() == 2 || (p->long_plan() == 0 && th::stft_subwave_default(p->g)));
    if (p->use_mel_mfma() && p->g.log2_nc >= 11 && p->d_mel_bsum != nullptr && p->kernel_choice != 7)
        return subwave ? "stft_subwave_kernel+mel_band_rows_kernel" : th::stft_is_block_plan(p->g) ? "stft_block_kernel+mel_band_rows_kernel" : "stft_wave_kernel+mel_band_rows_kernel";
    if (p->use_mel_mfma()) return subwave ? "stft_subwave_kernel+mel_mfma_kernel" : th::stft_is_block_plan(p->g) ? "stft_block_kernel+mel_mfma_kernel" : "stft_wave_kernel+mel_mfma_kernel";
    if (p->use_wave() && subwave) return "stft_subwave_kernel";
    if (p->use_wave() && th::stft_is_block_plan(p->g)) return "stft_block_kernel";
    return p->use_wave() ? "stft_wave_kernel" : p->bluestein() ? "stft_bluestein_kernel" : "stft_generic_kernel";
}

TH_API size_t th_pitch_f32(size_t n) { return (n + 31) / 32 * 32; }
TH_API size_t th_pitch_u16(size_t n) { return (n + 63) / 64 * 64; }

// ------------------------------------------------------------------------------------------ calc_spec
// d_range != NULL: also leave the global dB range [min_dB, max_dB] of these channels (core/mod.rs:169-180) in d_range —
// folded into the wave kernel's follow-up launch when the batch is one channel, else one more small launch
// Selector 11 (the in-order "sweep" chunk schedule, measurement apparatus): a wave whose ring slot never arrives gives its chunk
// up and raises word 1 of the queue block.  Read it back (4 bytes; blocks until the stream is idle) and fail the call that
// finds it: rows of that launch were left unwritten.
static int sweep_check(th_plan *p) {
    uint32_t flag = 0;
    TH_HIP(hipMemcpy(&flag, p->d_queue_head + 1, sizeof flag, hipMemcpyDeviceToHost));
    if (!flag) return TH_OK;
    TH_HIP(hipMemset(p->d_queue_head + 1, 0, sizeof flag));
    return fail(TH_ERR_INTERNAL, "sweep schedule (kernel selector 11): a wave gave up waiting for its chunk block; rows of the last launch are unwritten");
}

static int calc_spec_batch_impl(th_plan *p, const th_chan_desc *chans, size_t n_chan, float *d_minmax, float dB_range,
                                float *d_range) {
    TH_TRY
    TH_REQUIRE(p, "plan is NULL");
    TH_REQUIRE(d_range == nullptr || d_minmax != nullptr, "d_range needs d_minmax");
    if (n_chan == 0) return TH_OK;
    TH_REQUIRE(chans, "chans is NULL");
    TH_REQUIRE(n_chan < (1u << 24), "too many channels");
    th_ctx *c = p->ctx;
    const bool wave = p->use_wave();
    const bool mel_mfma = p->use_mel_mfma(), mel_fused = p->use_mel_fused();
    if (p->kernel_choice >= 2 && !wave)
        return fail(TH_ERR_UNSUPPORTED, "the wave kernels cover n_fft 512 .. 32768");
    StftGeom g = p->g;       // main launch
    StftGeom ge = p->g;      // edge launch (generic kernel)
    // wave kernel: chunk of consecutive frames one wave walks (the first frame of a chunk loads
    // n_fft samples, the rest only hop new ones).  32 amortises that well on large batches; small
    // batches (one track) get shorter chunks so that every wave of the chip has work.
    // interior frames [fa, fb) of a channel: the frame's whole n_fft-sample span [e0, e0 + n_fft), e0 = f*hop - win/2 -
    // pad_left, lies inside the channel (the wave kernel loads it unconditionally); the others are boundary frames
    // phased mode of the wave kernel (hop = 480-style framings): frames are loaded from the 128-sample grid below their
    // first window sample; chunks must start on frames that sit exactly on the grid (every fourth)
    // (not with the fused mel epilogue unless selector 5 asks for it: measured no gain there, with or without idle gaps —
    // 48 kHz / 347 mels 0.522 vs 0.521 ms, 44.1 kHz / 370 mels 0.513 vs 0.541 — the window then comes from LDS instead of
    // registers and the mel kernel is not bound by its loads)
    // (n_fft 4096 — the 40 ms default at 88.2 / 96 kHz — also with amplitude output for the matrix-core mel kernel)
    const int phase_mode = (wave && (!mel_mfma || g.log2_nc == 11) && (!mel_fused || p->kernel_choice == 5 || g.log2_nc == 11) && p->kernel_choice != 4 && p->d_wtab_phased != nullptr)
                               ? (mel_fused ? th::stft_wave_mel_phase_mode(g, p->wave_waves) : th::stft_wave_phased_mode(g, p->wave_waves)) : 0;
    const bool phased = phase_mode != 0;
    g.phased = (uint32_t)phase_mode;
    const int waves = p->wave_waves > 0 ? p->wave_waves : stft_wave_default_waves(g);
    // (multi-frame kernel with staged loads and hop % 4 == 2: see stft_wave_multi_tail_guard)
    const uint64_t tail_guard = (wave && !phased && th::stft_wave_multi_applies(g, mel_mfma ? 1 : (mel_fused ? 2 : 0)) && g.log2_nc == 8)
                                    ? th::stft_wave_multi_tail_guard(g) : 0;
    auto interior = [&g, phased, phase_mode, tail_guard](const th_chan_desc &d, uint64_t T, uint64_t &fa, uint64_t &fb) {
        if (phased) {
            const int64_t N = (int64_t)d.n_samples, half = (int64_t)(g.win / 2);
            auto delta = [&](int64_t f) { return (((f * (int64_t)g.hop - half) % 128) + 128) % 128; };
            auto start = [&](int64_t f) { return f * (int64_t)g.hop - half - delta(f); };
            int64_t a = 0, b = (int64_t)T;
            while (a < b && (start(a) < 0 || (phase_mode == 1 && delta(a) != 0))) a++;  // phased chunks start on the grid
            while (b > a && start(b - 1) + (int64_t)g.n_fft > N) b--;
            fa = (uint64_t)a;
            fb = (uint64_t)b;
            return;
        }
        const uint64_t lead = g.win / 2 + g.pad_left, N = d.n_samples > tail_guard ? d.n_samples - tail_guard : 0;
        fa = (lead + g.hop - 1) / g.hop;                                  // first f with e0 >= 0
        fb = N + lead >= g.n_fft ? (N + lead - g.n_fft) / g.hop + 1 : 0;  // one past the last f with e0 + n_fft <= N
        fa = std::min<uint64_t>(fa, T);
        fb = std::min<uint64_t>(std::max(fb, fa), T);
    };
    // boundary frames of channels with at least n_fft samples also go to the wave kernel (one-frame chunks with a
    // reflect-indexed fetch): no second launch on the fast path.  Not on the matrix-core mel path (its amplitude rows
    // are laid out for the interior jobs), not for channels shorter than n_fft (a single reflection is not enough there).
    const bool edges_in_wave = wave && !mel_mfma && !phased && !th::stft_is_block_plan(g);
    bool sweep = false;
    if (wave) {
        uint64_t total = 0;
        for (size_t i = 0; i < n_chan; i++) total += chans[i].n_frames;
        // about TH_CHUNKS_PER_WAVE chunks per wave (at most 32 frames each): the launch ends when the last chunk does,
        // so a chunk is the granularity of the load balance, while every chunk costs one full fetch and one atomic on the
        // queue head (served at ~8 ns each device-wide).  Measured inside bench.py's step: 4 per wave (29 frames)
        // 0.50-0.515 ms, 6 (19) 0.52, 8 (14) 0.54, 12 (9) 0.69; 32-frame chunks 0.53.
#if !defined(TH_CHUNKS_PER_WAVE)
#define TH_CHUNKS_PER_WAVE 4
#endif
        const uint64_t n_waves = (uint64_t)c->n_cu * (uint64_t)waves;
        uint64_t chunk;
        if (th::stft_is_block_plan(g) && g.log2_nc <= 14) {
            // Workgroup-per-frame plans (n_fft 8192 .. 32768): what runs side by side is one workgroup per CU (two at n_fft 8192),
            // not twelve waves, and a chunk start costs a workgroup launch + its set-up (30-50 us: tables / constants, the first
            // frame) or, in the persistent subwave kernel, a full reload of the resident samples.  About two chunks per slot, the
            // length that fills the last round best (round 5, profiles/r05_ab_block_chunks.txt: n_fft 16384 15 -> 44 frames 0.96 ->
            // 0.84-0.92 ms, 32768 8 -> 43 1.26 -> 1.21, 8192 30 -> 60-90 0.70 -> 0.64-0.69; n_fft 65536 measures 5 % slower with
            // longer chunks and keeps the rule below)
            const uint64_t slots = (uint64_t)c->n_cu * (g.log2_nc == 12 ? 2u : 1u);
            const uint64_t want = std::min<uint64_t>(128, std::max<uint64_t>(1, total / (slots * 2)));  // (a single track: one frame per chunk, every CU busy)
            uint64_t best = want, best_cost = ~0ull;
            for (uint64_t cand = std::max<uint64_t>(1, want - want / 4); cand <= want + want / 4; cand++) {
                uint64_t n = 0;
                for (size_t i = 0; i < n_chan; i++) {
                    uint64_t fa, fb;
                    interior(chans[i], chans[i].n_frames, fa, fb);
                    n += (fb - fa + cand - 1) / cand;
                }
                const uint64_t cost = (n + slots - 1) / slots * cand;
                if (cost < best_cost || (cost == best_cost && cand > best)) {
                    best_cost = cost;
                    best = cand;
                }
            }
            chunk = best;
        } else if (total <= n_waves * 32) {
            // small batch (one track ...): one chunk per wave, all of them assigned statically — the kernel then never
            // touches the queue (3072 waves finding it empty is 25 us of serialised atomics on a 20 us job)
            chunk = std::max<uint64_t>(1, (total + n_waves - 1) / n_waves);
            for (;; chunk++) {
                uint64_t n = 0;
                for (size_t i = 0; i < n_chan; i++) {
                    uint64_t fa, fb;
                    interior(chans[i], chans[i].n_frames, fa, fb);
                    n += (fb - fa + chunk - 1) / chunk;
                    if (edges_in_wave && chans[i].n_samples >= g.n_fft) n += chans[i].n_frames - (fb - fa);
                }
                if (n <= n_waves || chunk >= 32) break;
            }
        } else {
            // The launch ends when the last chunk does and every wave walks its chunks one after the other, so what counts
            // is rounds x chunk length: pick the chunk length (around total / (waves x TH_CHUNKS_PER_WAVE), at most 40
            // frames) whose chunk count fills the last round best.  Bench workload: 29 frames would be 12416 chunks = 4.04
            // rounds of 3072 waves, i.e. a fifth round for 128 stragglers: 0.520 ms; 30-32 frames 0.507 (r02_chunks).
            const uint64_t want = std::min<uint64_t>(32, std::max<uint64_t>(12, total / (n_waves * TH_CHUNKS_PER_WAVE)));
            uint64_t best = 0, best_cost = ~0ull;
            for (uint64_t cand = std::max<uint64_t>(12, want - want / 4); cand <= want + want / 4; cand++) {
                uint64_t n = 0;
                for (size_t i = 0; i < n_chan; i++) {
                    uint64_t fa, fb;
                    interior(chans[i], chans[i].n_frames, fa, fb);
                    n += (fb - fa + cand - 1) / cand;
                }
                const uint64_t cost = (n + n_waves - 1) / n_waves * cand;  // frames a wave walks through, at worst
                if (cost < best_cost || (cost == best_cost && cand > best)) {
                    best_cost = cost;
                    best = cand;
                }
            }
            chunk = best;
        }
        g.frames_per_tile = p->wave_chunk > 0 ? (uint32_t)p->wave_chunk : (uint32_t)chunk;
        // The sweep schedule (kernels_stft.hip: 4-frame chunks dealt out in order, next chunk prefetched) — round 4, selector 11
        // only: its memory skeleton streams 3-10 % faster (scripts/ubench/stft_skeleton.hip mode 8), the kernel does not: inside
        // bench.py's step 0.492 / 0.497 / 0.496 ms against 0.495 / 0.495 / 0.494 for the default schedule on one box, alternating,
        // and 8 % slower alone with 1 ms gaps (profiles/r04_ab_sweep.txt) — of which the kernel itself is 1 % (rocprofv3 trace:
        // 480 us against 475, profiles/r04_ab_sweep_kernel_level.txt); the rest is this function building and uploading a
        // seven times longer chunk table per call, which a pipelined step hides and a lone call does not.  With the packed-f32
        // pipeline (-39 % VALU instructions, +-0) that makes three independent changes of what the kernel DOES per frame that do
        // not change what a launch TAKES: it runs at the package power cap, where time follows flops + bytes, and neither moved.
        sweep = total > n_waves * 32 && p->wave_chunk == 0 && p->kernel_choice == 11 && !mel_mfma && !mel_fused &&
                g.n_mel == 0 && phase_mode == 0 && th::stft_wave_sweep_applies(g, p->wave_waves, 0);
        if (sweep) g.frames_per_tile = 4;
        // (an earlier launch of this plan that dropped a chunk: found here when the stream has gone idle — no stall otherwise —
        // and in th_calc_spec_host behind its synchronisation)
        if (sweep && hipStreamQuery(c->stream) == hipSuccess) {
            const int src = sweep_check(p);
            if (src != TH_OK) return src;
        }
        if (phase_mode == 1) g.frames_per_tile = std::max<uint32_t>(4, (g.frames_per_tile + 3) / 4 * 4);  // chunks start on the grid
    } else {
        g.frames_per_tile = 8;
        // (short transforms — n_fft 256 and below, the generic kernel's only full plans: a tile of 8 frames is one round of its frames-side-by-side
        // loop, and the workgroup's set-up and (min, max) fold then cost as much as the frames; up to 64 frames a tile where the batch still fills the chip)
        if (g.nc <= 256 && g.odd_m1 == 0) {
            uint64_t all = 0;
            for (size_t i = 0; i < n_chan; i++) all += chans[i].n_frames;
            const uint64_t want = all / ((uint64_t)c->n_cu * 16u);
            g.frames_per_tile = (uint32_t)std::min<uint64_t>(64, std::max<uint64_t>(8, want / 8 * 8));
        }
    }
    ge.frames_per_tile = 1;
    // main jobs: the wave kernel takes the interior frames [fa, fb) of every channel (all windowed
    // samples inside the signal); the generic kernel takes the boundary frames (reflect padding,
    // stft.rs:50-95) — or every frame when the wave kernel does not cover this plan.
    std::vector<ChanJob> jobs, edge;
    std::vector<uint32_t> tile_start, edge_start;
    uint64_t tiles = 0, edge_tiles = 0;
    // mel on the matrix cores: the wave kernel stores |X| rows (pitch amp_pitch) into plan-owned
    // scratch, mel_mfma_kernel contracts them with the filterbank into the caller's spec
    const uint32_t amp_pitch = (uint32_t)th_pitch_f32((size_t)p->mel_kblocks * 16);
    std::vector<MelJob> mel_jobs;
    std::vector<uint32_t> mel_start;
    std::vector<uint64_t> amp_row0(n_chan, 0);
    uint64_t mel_tiles = 0, amp_rows = 0;
    auto add = [](std::vector<ChanJob> &v, std::vector<uint32_t> &st, uint64_t &n_tiles, const StftGeom &gg,
                  const th_chan_desc &d, uint32_t T, uint32_t fb0, uint32_t fe0, uint32_t slot, bool wave_edge = false) {
        if (fb0 >= fe0) return;
        const uint32_t pitch = d.spec_pitch ? (uint32_t)d.spec_pitch : gg.height;
        v.push_back(ChanJob{d.wav, d.spec, (uint32_t)d.n_samples, T, fb0, fe0, slot, pitch, wave_edge ? 1u : 0u, 0u});
        st.push_back((uint32_t)n_tiles);
        const uint32_t fpt = wave_edge ? 1u : gg.frames_per_tile;
        n_tiles += (fe0 - fb0 + fpt - 1) / fpt;
    };
    for (size_t i = 0; i < n_chan; i++) {
        const th_chan_desc &d = chans[i];
        TH_REQUIRE(d.wav && d.spec, "channel %zu: NULL device pointer", i);
        TH_REQUIRE(d.n_samples >= 1 && d.n_samples < (1ull << 31) - 2 * g.n_fft, "channel %zu: n_samples=%llu out of range",
                   i, (unsigned long long)d.n_samples);
        const size_t T = stft_n_frames(d.n_samples, g.win, g.hop);
        TH_REQUIRE(d.n_frames == T, "channel %zu: n_frames=%llu but the framing gives %zu", i,
                   (unsigned long long)d.n_frames, T);
        TH_REQUIRE(d.spec_pitch == 0 || (d.spec_pitch >= g.height && d.spec_pitch < (1ull << 31)),
                   "channel %zu: spec_pitch %llu < height %u", i, (unsigned long long)d.spec_pitch, g.height);
        if (!wave) {
            add(jobs, tile_start, tiles, g, d, (uint32_t)T, 0, (uint32_t)T, (uint32_t)i);
        } else {
            uint64_t fa, fb;
            interior(d, T, fa, fb);
            add(jobs, tile_start, tiles, g, d, (uint32_t)T, (uint32_t)fa, (uint32_t)fb, (uint32_t)i);
            if (mel_mfma && fa < fb) {
                amp_row0[i] = amp_rows;
                amp_rows += T;
                mel_jobs.push_back(MelJob{nullptr, d.spec, (uint32_t)fa, (uint32_t)fb,
                                          (uint32_t)(d.spec_pitch ? d.spec_pitch : g.height), (uint32_t)i});
                mel_start.push_back((uint32_t)mel_tiles);
                mel_tiles += (fb - fa + MEL_TILE_FRAMES - 1) / MEL_TILE_FRAMES;
            }
            if (edges_in_wave && d.n_samples >= g.n_fft) {
                add(jobs, tile_start, tiles, g, d, (uint32_t)T, 0, (uint32_t)fa, (uint32_t)i, true);
                add(jobs, tile_start, tiles, g, d, (uint32_t)T, (uint32_t)fb, (uint32_t)T, (uint32_t)i, true);
            } else {
                add(edge, edge_start, edge_tiles, ge, d, (uint32_t)T, 0, (uint32_t)fa, (uint32_t)i);
                add(edge, edge_start, edge_tiles, ge, d, (uint32_t)T, (uint32_t)fb, (uint32_t)T, (uint32_t)i);
            }
        }
        TH_REQUIRE(tiles < (1ull << 31) && edge_tiles < (1ull << 31), "batch too large for one launch");
    }
    tile_start.push_back((uint32_t)tiles);
    edge_start.push_back((uint32_t)edge_tiles);

    mel_start.push_back((uint32_t)mel_tiles);
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    int rc = TH_OK;
    bool range_done = false;
    if (mel_mfma && amp_rows) {
        const size_t need = (size_t)amp_rows * amp_pitch * sizeof(float);
        if (need > p->amp_buf.cap || p->amp_zeroed < need) {
            TH_HIP(hipStreamSynchronize(c->stream));
            rc = p->amp_buf.ensure(need);
            if (rc != TH_OK) return rc;
            // the K tail (columns >= n_freq) must read as zero: the wave kernel never writes it
            TH_HIP(hipMemsetAsync(p->amp_buf.dptr, 0, p->amp_buf.cap, c->stream));
            p->amp_zeroed = p->amp_buf.cap;
        }
        float *amp = static_cast<float *>(p->amp_buf.dptr);
        size_t mj = 0;
        for (ChanJob &j : jobs) {  // interior jobs are in channel order, one per channel with interior frames
            j.spec = amp + amp_row0[j.mm_index] * amp_pitch;
            j.spec_pitch = amp_pitch;
            mel_jobs[mj++].amp = j.spec;
        }
        rc = p->mel_jobs.upload(c->stream, mel_jobs.data(), mel_jobs.size() * sizeof(MelJob));
        if (rc == TH_OK) rc = p->mel_tile_start.upload(c->stream, mel_start.data(), mel_start.size() * sizeof(uint32_t));
        if (rc != TH_OK) return rc;
    }
    rc = p->jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(ChanJob));
    if (wave) {
        // the wave kernels look a chunk up directly: (job, first frame) per chunk (cursor_at, kernels_stft.hip)
        std::vector<uint32_t> chunk_tab((size_t)tiles * 2);
        for (size_t j = 0; j < jobs.size(); j++) {
            const uint32_t fpt = jobs[j].edge ? 1u : g.frames_per_tile;
            for (uint32_t t = tile_start[j]; t < tile_start[j + 1]; t++) {
                chunk_tab[2 * (size_t)t] = (uint32_t)j;
                chunk_tab[2 * (size_t)t + 1] = jobs[j].f_begin + (t - tile_start[j]) * fpt;
            }
        }
        if (rc == TH_OK) rc = p->tile_start.upload(c->stream, chunk_tab.data(), chunk_tab.size() * sizeof(uint32_t));
    } else if (rc == TH_OK) {
        rc = p->tile_start.upload(c->stream, tile_start.data(), tile_start.size() * sizeof(uint32_t));
    }
    if (rc == TH_OK && !edge.empty()) {
        rc = p->edge_jobs.upload(c->stream, edge.data(), edge.size() * sizeof(ChanJob));
        if (rc == TH_OK) rc = p->edge_tile_start.upload(c->stream, edge_start.data(), edge_start.size() * sizeof(uint32_t));
    }
    if (rc != TH_OK) return rc;
    {   // n_fft >= 32768: the generic kernel's frame buffers live in global scratch (all frames of a plan without a fast
        // kernel, or the boundary frames of the n_fft 32768 block plan)
        const size_t need = !wave ? (p->bluestein() ? th::stft_bluestein_scratch_bytes(g, (uint32_t)tiles, c->n_cu) : th::stft_generic_scratch_bytes(g, (uint32_t)tiles, c->n_cu))
                                  : (edge.empty() ? 0 : th::stft_generic_scratch_bytes(ge, (uint32_t)edge_tiles, c->n_cu));
        if (need) {
            if (need > p->gen_scratch.cap) TH_HIP(hipStreamSynchronize(c->stream));  // (a launch may still use the old buffer)
            rc = p->gen_scratch.ensure(need);
            if (rc != TH_OK) return rc;
        }
    }
    // (min, max) slots: when every frame of every channel is in the wave launch, its last workgroup initialises and fills
    // them (no init launch); otherwise initialise here and let every kernel add with atomics
    const bool all_in_wave = wave && !mel_mfma && edge.empty() && tiles > 0;
    if (!all_in_wave) TH_HIP(launch_minmax_init(d_minmax, (uint32_t)n_chan, c->stream));
    // optional timing of the dominant kernel alone (th_plan_time_kernel): two events on the launch stream
    const bool timed = p->time_kernel && !p->ev_k0.empty();
    const size_t slot = (size_t)(p->timed_launches % th_plan::TIMER_SLOTS);
    if (timed) TH_HIP(hipEventRecord(p->ev_k0[slot], c->stream));
    if (wave) {
        th::WaveOut wo;
        wo.mode = mel_mfma ? 1 : (mel_fused ? 2 : 0);
        // n_fft 512, linear dB: four frames per wave (stft_wave_multi.h).  The same plan with two frames per wave exists for
        // n_fft 1024 (selector 6) but measures 0.72 ms against 0.64 for the one-frame plan on the bench tracks (it reloads
        // every frame in full: 16 loads per iteration against 4), so 1024 keeps the one-frame kernel by default.
        wo.multi = (th::stft_wave_multi_applies(g, wo.mode) && (g.log2_nc == 8 || p->kernel_choice == 6)) ? 1 : 0;
        wo.packed = p->kernel_choice == 9 ? 1 : 0;
        wo.sweep = sweep ? 1 : 0;
        wo.long_plan = p->long_plan();
        wo.subwave_twc = p->d_twc;
        if (mel_fused && (g.log2_nc >= 11 || p->use_mel_moment_small())) {  // moment form: the table stays in global memory (scalar + 16-byte lane loads)
            wo.mel_moment = g.log2_nc >= 11 ? 0u : 1u;
            wo.mel_tab = p->d_mel_mom;
            wo.mel_words = 0;
            wo.mel_slots = 0;
            wo.mel_groups = p->mel_mom_groups;
            wo.n_mel = g.n_mel;
        } else if (mel_fused && g.log2_nc == 8) {  // (the per-mel table of the banded sums, kernels.h)
            wo.mel_tab = p->d_mel_rows;
            wo.mel_groups = p->mel_rows_groups;
            wo.mel_words = p->mel_rows_groups * (uint32_t)(th::MEL_ROWS_W + 1) * 64u;
            wo.n_mel = g.n_mel;
        } else if (mel_fused && p->mel_bsum_fits()) {  // banded sums (mel_slots = 0 tells the kernel)
            wo.mel_tab = p->d_mel_bsum;
            wo.mel_words = p->mel_bsum_words;
            wo.mel_slots = 0;
            wo.mel_groups = p->mel_bsum_groups;
            for (uint32_t gq = 0; gq < 8; gq++) {
                wo.band_off[gq] = p->mel_bsum_hdr[2 * gq];
                wo.band_n[gq] = p->mel_bsum_hdr[2 * gq + 1];
            }
            wo.n_mel = g.n_mel;
            // frame pairs (round 5): two consecutive frames per pass over the table; selector 13 keeps the one-frame epilogue (A/B)
            wo.mel_pair = (p->kernel_choice != 13 && th::stft_wave_mel_pair_applies(g, p->wave_waves, p->mel_bsum_reach)) ? 1u : 0u;
        } else if (mel_fused) {
            wo.mel_tab = p->d_mel_fuse;
            wo.mel_words = p->mel_fuse_words;
            wo.mel_slots = p->mel_fuse_slots;
            wo.mel_groups = p->mel_fuse_groups;
            wo.n_mel = g.n_mel;
        }
        // min / max: the wave kernel stores one pair per chunk and its last workgroup folds them into the channel slots
        float *chunk_mm = nullptr;
        if (d_minmax != nullptr && wo.mode != 1 && tiles) {
            rc = p->chunk_mm.ensure((size_t)tiles * 2 * sizeof(float));
            if (rc != TH_OK) return rc;
            chunk_mm = static_cast<float *>(p->chunk_mm.dptr);
        }
        uint32_t n_post = 0;
        if (chunk_mm) {  // per-channel tile ranges for wave_post_kernel: a channel's jobs (interior, head, tail) are consecutive
            std::vector<th::WavePostJob> pj;
            for (size_t j = 0; j < jobs.size(); j++) {
                if (!pj.empty() && pj.back().mm_index == jobs[j].mm_index) pj.back().t1 = tile_start[j + 1];
                else pj.push_back(th::WavePostJob{tile_start[j], tile_start[j + 1], jobs[j].mm_index, 0u});
            }
            rc = p->post_jobs.upload(c->stream, pj.data(), pj.size() * sizeof(th::WavePostJob));
            if (rc != TH_OK) return rc;
            n_post = (uint32_t)pj.size();
        }
        // The chunk queue head is rewound by wave_post_kernel right behind the wave kernel.  Should anything between the
        // two launches fail (the event record, the post launch itself), queue_dirty stays set and the next launch zeroes
        // the head itself — a stale head would make it skip chunks and leave spectrogram rows unwritten without an error.
        if (p->queue_dirty) TH_HIP(hipMemsetAsync(p->d_queue_head, 0, sizeof(uint32_t), c->stream));
        p->queue_dirty = true;
        TH_HIP(launch_stft_wave(g, (const ChanJob *)p->jobs.dptr, (const uint32_t *)p->tile_start.dptr,
                                (uint32_t)jobs.size(), (uint32_t)tiles, phased ? p->d_wtab_phased : p->d_wtab, p->d_tw, chunk_mm, p->d_queue_head,
                                c->n_cu, waves, wo, c->stream));
        if (timed) TH_HIP(hipEventRecord(p->ev_k1[slot], c->stream));
        // fold the per-chunk (min, max) pairs into the channel slots and rewind the chunk queue
        range_done = d_range != nullptr && all_in_wave && n_post == 1 && n_chan == 1;
        TH_HIP(launch_wave_post((const th::WavePostJob *)p->post_jobs.dptr, n_post, chunk_mm, d_minmax, all_in_wave,
                                p->d_queue_head, dB_range, range_done ? d_range : nullptr, c->stream));
        p->queue_dirty = false;
        if (mel_mfma && p->d_mel_rows != nullptr && p->kernel_choice != 7)
            TH_HIP(launch_mel_rows((const MelJob *)p->mel_jobs.dptr, (const uint32_t *)p->mel_tile_start.dptr,
                                   (uint32_t)mel_jobs.size(), (uint32_t)mel_tiles, amp_pitch, p->d_mel_rows, p->mel_rows_groups,
                                   g.n_mel, d_minmax, c->n_cu, c->stream));
        else if (mel_mfma && g.log2_nc >= 11 && p->d_mel_bsum != nullptr && p->kernel_choice != 7)
            TH_HIP(launch_mel_band_rows((const MelJob *)p->mel_jobs.dptr, (const uint32_t *)p->mel_tile_start.dptr,
                                        (uint32_t)mel_jobs.size(), (uint32_t)mel_tiles, amp_pitch, g.n_freq, p->d_mel_bsum,
                                        p->mel_bsum_words, p->mel_bsum_groups, p->mel_bsum_hdr, g.n_mel, d_minmax, c->n_cu,
                                        c->stream));
        else if (mel_mfma)
            TH_HIP(launch_mel_mfma((const MelJob *)p->mel_jobs.dptr, (const uint32_t *)p->mel_tile_start.dptr,
                                   (uint32_t)mel_jobs.size(), (uint32_t)mel_tiles, amp_pitch, p->d_mel_bt, p->d_mel_band,
                                   p->d_mel_slice, p->mel_slices, p->mel_zero_block, g.n_mel, d_minmax, c->stream));
        if (!edge.empty())  // boundary frames the wave kernel did not take: generic kernel (reflect padding; mel included)
            TH_HIP(launch_stft_generic(ge, (const ChanJob *)p->edge_jobs.dptr, (const uint32_t *)p->edge_tile_start.dptr,
                                       (uint32_t)edge.size(), (uint32_t)edge_tiles, p->d_window, p->d_tw, p->d_mel_fb,
                                       p->d_mel_lo, p->d_mel_hi, d_minmax, c->stream, p->gen_scratch.dptr, c->n_cu));
    } else {
        if (p->bluestein())
            TH_HIP(launch_stft_bluestein(g, (const ChanJob *)p->jobs.dptr, (const uint32_t *)p->tile_start.dptr, (uint32_t)jobs.size(),
                                         (uint32_t)tiles, p->d_window, p->d_bs_chirp, p->d_bs_bhat, p->d_bs_twm, p->d_bs_tws, p->d_mel_fb,
                                         p->d_mel_lo, p->d_mel_hi, d_minmax, c->stream, p->gen_scratch.dptr, c->n_cu));
        else
        TH_HIP(launch_stft_generic(g, (const ChanJob *)p->jobs.dptr, (const uint32_t *)p->tile_start.dptr,
                                   (uint32_t)jobs.size(), (uint32_t)tiles, p->d_window, p->d_tw, p->d_mel_fb,
                                   p->d_mel_lo, p->d_mel_hi, d_minmax, c->stream, p->gen_scratch.dptr, c->n_cu));
        if (timed) TH_HIP(hipEventRecord(p->ev_k1[slot], c->stream));
    }
    if (timed) p->timed_launches++;
    if (d_range != nullptr && !range_done)
        TH_HIP(launch_minmax_reduce(d_minmax, (uint32_t)n_chan, nullptr, dB_range, d_range, c->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_calc_spec_batch_dev(th_plan *p, const th_chan_desc *chans, size_t n_chan, float *d_minmax) {
    return calc_spec_batch_impl(p, chans, n_chan, d_minmax, 0.0f, nullptr);
}

TH_API int th_calc_spec_batch_ranged_dev(th_plan *p, const th_chan_desc *chans, size_t n_chan, float *d_minmax, float dB_range,
                                         float *d_range) {
    if (!d_range) return fail(TH_ERR_INVALID_ARG, "d_range is NULL");
    return calc_spec_batch_impl(p, chans, n_chan, d_minmax, dB_range, d_range);
}

TH_API int th_minmax_reduce_dev(th_ctx *c, const float *d_minmax, size_t n_chan, float *d_out) {
    TH_TRY
    TH_REQUIRE(c && d_out, "NULL argument");
    TH_REQUIRE(n_chan == 0 || d_minmax, "d_minmax is NULL");
    TH_REQUIRE(n_chan < (1ull << 31), "too many channels");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(launch_minmax_reduce(d_minmax, (uint32_t)n_chan, d_out, 0.0f, nullptr, c->stream));
    return TH_OK;
    TH_CATCH
}

// single-GPU form of th_minmax_reduce_dev + th_global_db_range_dev: one launch instead of two
TH_API int th_minmax_reduce_range_dev(th_ctx *c, const float *d_minmax, size_t n_chan, float dB_range, float *d_min_negmax,
                                      float *d_range) {
    TH_TRY
    TH_REQUIRE(c && d_range, "NULL argument");
    TH_REQUIRE(n_chan == 0 || d_minmax, "d_minmax is NULL");
    TH_REQUIRE(n_chan < (1ull << 31), "too many channels");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(launch_minmax_reduce(d_minmax, (uint32_t)n_chan, d_min_negmax, dB_range, d_range, c->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_plan_time_kernel(th_plan *p, int enable) {
    TH_TRY
    TH_REQUIRE(p, "plan is NULL");
    std::lock_guard<std::recursive_mutex> lk(p->ctx->mu);
    TH_HIP(hipSetDevice(p->ctx->device));
    if (enable && p->ev_k0.empty()) {
        for (size_t i = 0; i < th_plan::TIMER_SLOTS; i++) {
            hipEvent_t a = nullptr, b = nullptr;
            TH_HIP(hipEventCreate(&a));
            p->ev_k0.push_back(a);
            TH_HIP(hipEventCreate(&b));
            p->ev_k1.push_back(b);
        }
    }
    p->time_kernel = enable != 0;
    p->timed_launches = 0;
    return TH_OK;
    TH_CATCH
}

TH_API int th_plan_kernel_ms_history(th_plan *p, float *out_ms, size_t capacity, size_t *n_out) {
    TH_TRY
    TH_REQUIRE(p && n_out && (out_ms || capacity == 0), "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(p->ctx->mu);
    const size_t have = (size_t)std::min<uint64_t>(p->timed_launches, th_plan::TIMER_SLOTS);
    const size_t n = std::min(have, capacity);
    *n_out = n;
    for (size_t i = 0; i < n; i++) {  // oldest kept launch first
        const size_t slot = (size_t)((p->timed_launches - n + i) % th_plan::TIMER_SLOTS);
        TH_HIP(hipEventSynchronize(p->ev_k1[slot]));
        TH_HIP(hipEventElapsedTime(&out_ms[i], p->ev_k0[slot], p->ev_k1[slot]));
    }
    // (selector 11, A/B builds: a pipelined caller never lets the stream go idle between launches, so the launch path's check for a
    // dropped chunk never runs there — this call has just waited for the launches it reports, check here as well: ADVICE r5)
    if (n > 0 && p->kernel_choice == 11) {
        TH_HIP(hipStreamSynchronize(p->ctx->stream));
        const int src = sweep_check(p);
        if (src != TH_OK) return src;
    }
    return TH_OK;
    TH_CATCH
}

TH_API int th_plan_last_kernel_ms(th_plan *p, float *ms) {
    TH_TRY
    TH_REQUIRE(p && ms, "NULL argument");
    size_t n = 0;
    const int rc = th_plan_kernel_ms_history(p, ms, 1, &n);
    if (rc != TH_OK) return rc;
    TH_REQUIRE(n == 1, "no timed launch: call th_plan_time_kernel(plan, 1) and th_calc_spec_batch_dev first");
    return TH_OK;
    TH_CATCH
}

TH_API int th_calc_spec_host(th_plan *p, const float *wav, size_t n_samples, float *out_spec, size_t cap,
                             size_t *n_frames, float *out_min, float *out_max) {
    TH_TRY
    TH_REQUIRE(p && wav && out_spec && n_frames, "NULL argument");
    TH_REQUIRE(n_samples >= 1, "empty input");
    const StftGeom &g = p->g;
    const size_t T = stft_n_frames(n_samples, g.win, g.hop);
    *n_frames = T;
    if ((size_t)T * g.height > cap) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu floats", (size_t)T * g.height);
    th_ctx *c = p->ctx;
    TH_HIP(hipSetDevice(c->device));
    float *d_wav = nullptr, *d_spec = nullptr, *d_mm = nullptr;
    int rc = TH_OK;
    auto cleanup = [&]() {
        if (d_wav) (void)hipFree(d_wav);
        if (d_spec) (void)hipFree(d_spec);
        if (d_mm) (void)hipFree(d_mm);
    };
    auto hipok = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && rc == TH_OK) rc = fail(e == hipErrorOutOfMemory ? TH_ERR_OOM : TH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
        return e == hipSuccess;
    };
    if (hipok(hipMalloc((void **)&d_wav, n_samples * sizeof(float)), "hipMalloc wav") &&
        hipok(hipMalloc((void **)&d_spec, std::max<size_t>(1, T * g.height) * sizeof(float)), "hipMalloc spec") &&
        hipok(hipMalloc((void **)&d_mm, 2 * sizeof(float)), "hipMalloc minmax") &&
        hipok(hipMemcpy(d_wav, wav, n_samples * sizeof(float), hipMemcpyHostToDevice), "upload")) {
        th_chan_desc d{d_wav, d_spec, n_samples, T, 0};
        rc = th_calc_spec_batch_dev(p, &d, 1, d_mm);
        if (rc == TH_OK) {
            float mm[2];
            if (hipok(hipStreamSynchronize(c->stream), "sync") && (p->kernel_choice != 11 || (rc = sweep_check(p)) == TH_OK) &&
                hipok(hipMemcpy(out_spec, d_spec, T * g.height * sizeof(float), hipMemcpyDeviceToHost), "download") &&
                hipok(hipMemcpy(mm, d_mm, sizeof mm, hipMemcpyDeviceToHost), "download minmax")) {
                if (out_min) *out_min = mm[0];
                if (out_max) *out_max = mm[1];
            }
        }
    }
    cleanup();
    return rc;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ spec → img
// shared body: host range (d_range == nullptr) or device-resident range [min_dB, max_dB]
static int spec_to_img_impl(th_ctx *c, const th_img_desc *descs, size_t n, float min_dB, float max_dB, const float *d_range,
                            uint32_t colormap_len) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (n == 0) return TH_OK;
    TH_REQUIRE(descs, "descs is NULL");
    const bool all_neg_inf = !d_range && (min_dB == max_dB) && std::isinf(max_dB) && max_dB < 0;  // drawing.rs:16-18
    if (!all_neg_inf && !d_range) TH_REQUIRE(std::isfinite(min_dB), "min_dB must be finite (drawing.rs:19)");
    if (!all_neg_inf) {
        // same batch as the previous call (re-quantise after a dB-range / colormap change, benchmark loops): the
        // device tables are still valid, skip validation and table building
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        if (c->img_descs_key.size() == n * sizeof(th_img_desc) && std::memcmp(c->img_descs_key.data(), descs, n * sizeof(th_img_desc)) == 0 &&
            c->img_jobs.dptr && c->img_start.dptr) {
            TH_HIP(hipSetDevice(c->device));
            TH_HIP(launch_spec_to_img((const ImgJob *)c->img_jobs.dptr, (const uint32_t *)c->img_start.dptr, (uint32_t)n,
                                      c->img_tiles_key, min_dB, max_dB, colormap_len, d_range, c->stream));
            return TH_OK;
        }
    }
    std::vector<ImgJob> jobs(n);
    std::vector<uint32_t> start;  // job index of every block
    uint64_t tiles = 0;
    for (size_t i = 0; i < n; i++) {
        const th_img_desc &d = descs[i];
        TH_REQUIRE(d.i_end >= d.i_start, "desc %zu: i_end < i_start", i);
        const uint64_t out_h = d.i_end - d.i_start;
        TH_REQUIRE(d.n_frames < (1ull << 31) && d.height < (1ull << 31) && d.i_end < (1ull << 31), "desc %zu: too large", i);
        TH_REQUIRE((d.spec && d.img) || out_h * d.n_frames == 0, "desc %zu: NULL device pointer", i);
        TH_REQUIRE(d.spec_pitch == 0 || (d.spec_pitch >= d.height && d.spec_pitch < (1ull << 31)), "desc %zu: bad spec_pitch", i);
        TH_REQUIRE(d.img_pitch == 0 || (d.img_pitch >= d.n_frames && d.img_pitch < (1ull << 31)), "desc %zu: bad img_pitch", i);
        const uint64_t nt = ((d.n_frames + IMG_TILE_T - 1) / IMG_TILE_T) * ((out_h + IMG_TILE_F - 1) / IMG_TILE_F);
        TH_REQUIRE(tiles + nt < (1ull << 27), "batch too large for one launch");
        jobs[i] = ImgJob{d.spec, d.img, (uint32_t)d.n_frames, (uint32_t)d.height, (uint32_t)d.i_start, (uint32_t)d.i_end,
                         (uint32_t)(d.spec_pitch ? d.spec_pitch : d.height), (uint32_t)(d.img_pitch ? d.img_pitch : d.n_frames),
                         (uint32_t)tiles, (uint32_t)nt};
        tiles += nt;
        start.insert(start.end(), (size_t)nt, (uint32_t)i);  // block -> job table
    }
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    if (all_neg_inf) {
        for (size_t i = 0; i < n; i++) {
            const size_t rows = descs[i].i_end - descs[i].i_start, pitch = jobs[i].img_pitch;
            if (rows && descs[i].n_frames)
                TH_HIP(hipMemset2DAsync(descs[i].img, pitch * sizeof(uint16_t), 0, descs[i].n_frames * sizeof(uint16_t), rows,
                                        c->stream));
        }
        return TH_OK;
    }
    c->img_descs_key.clear();   // the tables are about to be overwritten: the key names them only once BOTH uploads succeeded
    int rc = c->img_jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(ImgJob));
    if (rc != TH_OK) return rc;
    rc = c->img_start.upload(c->stream, start.data(), start.size() * sizeof(uint32_t));
    if (rc != TH_OK) return rc;
    c->img_descs_key.assign(reinterpret_cast<const unsigned char *>(descs), reinterpret_cast<const unsigned char *>(descs + n));
    c->img_tiles_key = (uint32_t)tiles;
    TH_HIP(launch_spec_to_img((const ImgJob *)c->img_jobs.dptr, (const uint32_t *)c->img_start.dptr, (uint32_t)n,
                              (uint32_t)tiles, min_dB, max_dB, colormap_len, d_range, c->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_spec_to_img_batch_dev(th_ctx *c, const th_img_desc *descs, size_t n, float min_dB, float max_dB,
                                    uint32_t colormap_len) {
    return spec_to_img_impl(c, descs, n, min_dB, max_dB, nullptr, colormap_len);
}

// the dB range stays on the device (th_minmax_reduce_dev -> [all-reduce] -> th_global_db_range_dev): no host round
// trip between the STFT stage and the quantiser
TH_API int th_spec_to_img_batch_dev_ranged(th_ctx *c, const th_img_desc *descs, size_t n, const float *d_range,
                                           uint32_t colormap_len) {
    if (!d_range) return fail(TH_ERR_INVALID_ARG, "d_range is NULL");
    return spec_to_img_impl(c, descs, n, 0.0f, 0.0f, d_range, colormap_len);
}

// Quantise + level-0 raster of every tile in one pass (kernels_image.hip: spec_to_img_raster_kernel)
TH_API int th_spec_to_img_raster_batch_dev(th_ctx *c, const th_img_tiles_desc *descs, size_t n, float min_dB, float max_dB,
                                           const float *d_range, const uint8_t *d_colormap, uint32_t n_colors) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (n == 0) return TH_OK;
    TH_REQUIRE(descs && d_colormap && n_colors >= 1, "NULL descs / colormap or empty colormap");
    TH_REQUIRE(n_colors <= 65536, "colormaps of more than 65536 entries are not supported");
    const bool all_neg_inf = !d_range && (min_dB == max_dB) && std::isinf(max_dB) && max_dB < 0;  // drawing.rs:16-18
    if (!all_neg_inf && !d_range) TH_REQUIRE(std::isfinite(min_dB), "min_dB must be finite (drawing.rs:19)");
    // key of the batch: the descriptors and every tile pointer (identical batch -> the device tables are reused as they are)
    std::vector<unsigned char> key;
    size_t n_ptrs = 0;
    for (size_t i = 0; i < n; i++) n_ptrs += (size_t)descs[i].n_tiles_x * descs[i].n_tiles_y;
    key.reserve(n * sizeof(th_img_tiles_desc) + n_ptrs * sizeof(void *));
    for (size_t i = 0; i < n; i++) {
        const th_img_tiles_desc &d = descs[i];
        TH_REQUIRE(d.tiles || (size_t)d.n_tiles_x * d.n_tiles_y == 0, "desc %zu: tiles is NULL", i);
        const unsigned char *p = reinterpret_cast<const unsigned char *>(&d);
        key.insert(key.end(), p, p + sizeof(th_img_tiles_desc));
        const unsigned char *q = reinterpret_cast<const unsigned char *>(d.tiles);
        key.insert(key.end(), q, q + (size_t)d.n_tiles_x * d.n_tiles_y * sizeof(void *));
    }
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    if (!(c->fused_key == key && c->fused_jobs.dptr && c->fused_start.dptr && c->fused_ptrs.dptr)) {
        std::vector<FusedJob> jobs(n);
        std::vector<uint32_t> start;
        std::vector<uint8_t *> ptrs;
        ptrs.reserve(n_ptrs);
        uint64_t blocks = 0;
        for (size_t i = 0; i < n; i++) {
            const th_img_desc &d = descs[i].img;
            TH_REQUIRE(d.i_end >= d.i_start, "desc %zu: i_end < i_start", i);
            const uint64_t out_h = d.i_end - d.i_start;
            TH_REQUIRE(d.n_frames < (1ull << 31) && d.height < (1ull << 31) && d.i_end < (1ull << 31), "desc %zu: too large", i);
            TH_REQUIRE((d.spec && d.img && d.height >= 1) || out_h * d.n_frames == 0, "desc %zu: NULL device pointer or empty spec", i);
            TH_REQUIRE(d.spec_pitch == 0 || (d.spec_pitch >= d.height && d.spec_pitch < (1ull << 31)), "desc %zu: bad spec_pitch", i);
            TH_REQUIRE(d.img_pitch == 0 || (d.img_pitch >= d.n_frames && d.img_pitch < (1ull << 31)), "desc %zu: bad img_pitch", i);
            const uint64_t n_tx = out_h && d.n_frames ? (d.n_frames + 511) / 512 : 0, n_ty = out_h && d.n_frames ? (out_h + 511) / 512 : 0;
            TH_REQUIRE(descs[i].n_tiles_x == n_tx && descs[i].n_tiles_y == n_ty, "desc %zu: the image has %llu x %llu level-0 tiles, not %u x %u",
                       i, (unsigned long long)n_tx, (unsigned long long)n_ty, descs[i].n_tiles_x, descs[i].n_tiles_y);
            const uint64_t n_bands = (out_h + FUSED_FB - 1) / FUSED_FB, nb = n_tx * n_bands;
            TH_REQUIRE(blocks + nb < (1ull << 27) && ptrs.size() + n_tx * n_ty < (1ull << 31), "batch too large for one launch");
            for (uint64_t t = 0; t < n_tx * n_ty; t++) {
                uint8_t *tp = descs[i].tiles[t];
                TH_REQUIRE((reinterpret_cast<uintptr_t>(tp) & 3u) == 0, "desc %zu: tile %llu must be 4-byte aligned", i, (unsigned long long)t);
                ptrs.push_back(tp);
            }
            jobs[i] = FusedJob{d.spec, d.img, (uint32_t)d.n_frames, (uint32_t)d.height, (uint32_t)d.i_start, (uint32_t)d.i_end,
                               (uint32_t)(d.spec_pitch ? d.spec_pitch : d.height), (uint32_t)(d.img_pitch ? d.img_pitch : d.n_frames),
                               (uint32_t)blocks, (uint32_t)std::max<uint64_t>(n_bands, 1), (uint32_t)n_tx, (uint32_t)n_ty,
                               (uint32_t)(ptrs.size() - n_tx * n_ty), 0u};
            blocks += nb;
            start.insert(start.end(), (size_t)nb, (uint32_t)i);
        }
        if (ptrs.empty()) ptrs.push_back(nullptr);
        c->fused_key.clear();   // the previous batch's key must not survive a partly failed re-upload of the three tables
        int rc = c->fused_jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(FusedJob));
        if (rc == TH_OK && !start.empty()) rc = c->fused_start.upload(c->stream, start.data(), start.size() * sizeof(uint32_t));
        if (rc == TH_OK) rc = c->fused_ptrs.upload(c->stream, ptrs.data(), ptrs.size() * sizeof(uint8_t *));
        if (rc != TH_OK) return rc;
        c->fused_key.swap(key);
        c->fused_blocks_key = (uint32_t)blocks;
    }
    TH_HIP(launch_spec_to_img_raster((const FusedJob *)c->fused_jobs.dptr, (const uint32_t *)c->fused_start.dptr, c->fused_blocks_key,
                                     (uint8_t *const *)c->fused_ptrs.dptr, min_dB, max_dB, d_range, all_neg_inf ? 1 : 0, d_colormap,
                                     n_colors, c->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_global_db_range_dev(th_ctx *c, const float *d_min_negmax, float dB_range, float *d_range) {
    TH_TRY
    TH_REQUIRE(c && d_min_negmax && d_range, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(launch_db_range(d_min_negmax, dB_range, d_range, c->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_spec_to_img_dev(th_ctx *c, const float *d_spec, size_t n_frames, size_t height, size_t i_start,
                              size_t i_end, float min_dB, float max_dB, uint32_t colormap_len, uint16_t *d_img) {
    th_img_desc d{d_spec, d_img, n_frames, height, i_start, i_end, 0, 0};
    return th_spec_to_img_batch_dev(c, &d, 1, min_dB, max_dB, colormap_len);
}

// ------------------------------------------------------------------------------------------ raster
TH_API int th_raster_tiles_dev(th_ctx *c, const th_raster_desc *descs, size_t n, const uint8_t *d_colormap,
                               uint32_t n_colors) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (n == 0) return TH_OK;
    TH_REQUIRE(descs && d_colormap && n_colors >= 1, "NULL descs/colormap or empty colormap");
    TH_REQUIRE(n_colors <= 65536, "colormaps of more than 65536 entries are not supported");
    {   // same batch as the previous call: the device tables are still valid
        std::lock_guard<std::recursive_mutex> lk(c->mu);
        if (c->raster_descs_key.size() == n * sizeof(th_raster_desc) &&
            std::memcmp(c->raster_descs_key.data(), descs, n * sizeof(th_raster_desc)) == 0 && c->raster_jobs.dptr && c->raster_start.dptr) {
            TH_HIP(hipSetDevice(c->device));
            TH_HIP(launch_raster_level0((const RasterJob *)c->raster_jobs.dptr, (const uint32_t *)c->raster_start.dptr,
                                        (uint32_t)n, c->raster_blocks_key, d_colormap, n_colors, c->stream));
            return TH_OK;
        }
    }
    std::vector<RasterJob> jobs(n);
    std::vector<uint32_t> start;  // job index of every block
    uint64_t blocks = 0;
    for (size_t i = 0; i < n; i++) {
        const th_raster_desc &d = descs[i];
        TH_REQUIRE((uint64_t)d.origin_x + d.width <= d.img_width && (uint64_t)d.origin_y + d.height <= d.img_height,
                   "desc %zu: tile rectangle outside the image", i);
        const uint64_t px = (uint64_t)d.width * d.height;
        TH_REQUIRE(px < (1ull << 31), "desc %zu: tile too large", i);
        TH_REQUIRE(px == 0 || (d.img && d.rgba), "desc %zu: NULL device pointer", i);
        TH_REQUIRE((reinterpret_cast<uintptr_t>(d.rgba) & 3u) == 0, "desc %zu: rgba must be 4-byte aligned", i);
        TH_REQUIRE(d.img_pitch == 0 || d.img_pitch >= d.img_width, "desc %zu: img_pitch < img_width", i);
        const uint32_t qpr = (d.width + 3) / 4;
        const uint32_t inv = qpr > 1 ? (uint32_t)((1ull << 32) / qpr) + 1u : 0u;  // exact for q * qpr < 2^32
        TH_REQUIRE(px == 0 || (px + 4) * d.width < (1ull << 32), "desc %zu: tile too large", i);
        const uint32_t inv_w = d.width > 1 ? (uint32_t)((1ull << 32) / d.width) + 1u : 0u;
        // (+1: a tile base off the 16-byte grid shifts the quads by up to 3 pixels, see raster_quads)
        const uint64_t nb = ((uint64_t)qpr * d.height + 1 + RASTER_QUADS_PER_BLOCK - 1) / RASTER_QUADS_PER_BLOCK;
        TH_REQUIRE(blocks + nb < (1ull << 27), "batch too large for one launch");
        jobs[i] = RasterJob{d.img, d.rgba, d.img_width, d.img_height, d.origin_x, d.origin_y, d.width, d.height,
                            d.img_pitch ? d.img_pitch : d.img_width, qpr, inv, inv_w, (uint32_t)blocks};
        blocks += nb;
        start.insert(start.end(), (size_t)nb, (uint32_t)i);  // block -> job table
    }
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    c->raster_descs_key.clear();   // (as above: a failed second upload must not leave the old key on mixed tables)
    int rc = c->raster_jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(RasterJob));
    if (rc != TH_OK) return rc;
    rc = c->raster_start.upload(c->stream, start.data(), start.size() * sizeof(uint32_t));
    if (rc != TH_OK) return rc;
    c->raster_descs_key.assign(reinterpret_cast<const unsigned char *>(descs), reinterpret_cast<const unsigned char *>(descs + n));
    c->raster_blocks_key = (uint32_t)blocks;
    TH_HIP(launch_raster_level0((const RasterJob *)c->raster_jobs.dptr, (const uint32_t *)c->raster_start.dptr,
                                (uint32_t)n, (uint32_t)blocks, d_colormap, n_colors, c->stream));
    return TH_OK;
    TH_CATCH
}

static void put_u32(uint8_t *p, uint32_t v) { std::memcpy(p, &v, 4); }  // little-endian host (x86-64)
static void put_u64(uint8_t *p, uint64_t v) { std::memcpy(p, &v, 8); }

TH_API int th_encode_spectrogram_tile_dev(th_ctx *c, const uint16_t *d_img, size_t img_height, size_t img_width,
                                          size_t img_pitch, const uint8_t *colormap_rgba, size_t colormap_bytes, uint64_t revision,
                                          uint32_t level_x, uint32_t level_y, uint32_t tile_x, uint32_t tile_y,
                                          uint8_t *out, size_t cap, size_t *out_len) {
    TH_TRY
    TH_REQUIRE(c && out && out_len, "NULL argument");
    TH_REQUIRE(colormap_rgba && colormap_bytes >= 4 && colormap_bytes % 4 == 0, "colormap must be a non-empty RGBA8 array");
    TH_REQUIRE(img_width < (1ull << 31) && img_height < (1ull << 31), "image too large");
    TH_REQUIRE(img_pitch == 0 || (img_pitch >= img_width && img_pitch < (1ull << 31)), "img_pitch < img_width");
    const TileGeom g = spectrogram_tile_geometry(img_width, img_height, level_x, level_y, tile_x, tile_y);
    const size_t need = 40 + g.width * g.height * 4;
    *out_len = need;
    if (cap < need) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", need);
    put_u64(out, revision);
    put_u32(out + 8, (uint32_t)g.width);
    put_u32(out + 12, (uint32_t)g.height);
    put_u32(out + 16, level_x);
    put_u32(out + 20, level_y);
    put_u32(out + 24, tile_x);
    put_u32(out + 28, tile_y);
    put_u32(out + 32, (uint32_t)g.origin_x);
    put_u32(out + 36, (uint32_t)g.origin_y);
    if (g.width == 0 || g.height == 0) return TH_OK;
    TH_REQUIRE(d_img, "d_img is NULL");
    const uint32_t n_colors = (uint32_t)(colormap_bytes / 4);
    std::lock_guard<std::recursive_mutex> lk(c->mu);  // tile_out / colormap / lod scratch: whole request
    TH_HIP(hipSetDevice(c->device));
    int rc = c->colormap.upload(c->stream, colormap_rgba, colormap_bytes);
    if (rc != TH_OK) return rc;
    rc = c->tile_out.ensure(g.width * g.height * 4);
    if (rc != TH_OK) return rc;
    const uint32_t pitch = (uint32_t)(img_pitch ? img_pitch : img_width);
    if (level_x == 0 && level_y == 0) {
        // integral crop box equal to the destination size: exact copy (pinned by render_tiles.rs:464-471)
        th_raster_desc d{d_img, (uint8_t *)c->tile_out.dptr, (uint32_t)img_width, (uint32_t)img_height,
                         (uint32_t)g.origin_x, (uint32_t)g.origin_y, (uint32_t)g.width, (uint32_t)g.height, pitch, 0};
        rc = th_raster_tiles_dev(c, &d, 1, (const uint8_t *)c->colormap.dptr, n_colors);
        if (rc != TH_OK) return rc;
    } else {
        // LOD > 0: separable Lanczos3 of the crop box (render_tiles.rs:354-393), then the same raster.
        // Tap tables are built here in f64 exactly as the CPU restatement does, in the arithmetic of Pillow's
        // ImagingResample (bit-identical to Pillow on the committed fixtures; formally unpinned against
        // fast_image_resize itself).
        const size_t dw = g.width, dh = g.height;
        const double W = (double)img_width, Hh = (double)img_height;
        const double left = (double)g.origin_x * W / (double)g.lod_w, top = (double)g.origin_y * Hh / (double)g.lod_h;
        const double cw = (double)(g.origin_x + dw) * W / (double)g.lod_w - left;
        const double chh = (double)(g.origin_y + dh) * Hh / (double)g.lod_h - top;
        const double scy = chh / (double)dh, fy = scy < 1.0 ? 1.0 : scy, supy = 3.0 * fy;
        long y_lo = (long)std::floor(top - supy) - 1, y_hi = (long)std::ceil(top + chh + supy) + 1;
        if (y_lo < 0) y_lo = 0;
        if (y_hi > (long)img_height) y_hi = (long)img_height;
        const size_t n_rows = (size_t)(y_hi - y_lo);
        // a tap table beyond this is a level no viewer asks for; refuse instead of allocating GBs
        const double est_taps = 6.0 * std::max(cw / (double)dw, chh / (double)dh) + 4.0;
        if (est_taps * 8.0 * (double)std::max(dw, dh) > 256.0 * 1024 * 1024)
            return fail(TH_ERR_UNSUPPORTED, "LOD level (%u,%u) needs a tap table beyond 256 MB", level_x, level_y);
        LodAxisHost ax, ay;
        build_lod_axis(left, cw, dw, 0, (long)img_width, ax);
        build_lod_axis(top, chh, dh, y_lo, y_hi, ay);
        // one blob: [x: start,count,wsum,w][y: start,count,wsum,w], 8-byte aligned sections
        const size_t bx = ax.blob_bytes(dw), by = ay.blob_bytes(dh);
        std::vector<unsigned char> blob(bx + by);
        ax.pack(blob.data(), dw);
        ay.pack(blob.data() + bx, dh);
        rc = c->lod_tabs.upload(c->stream, blob.data(), blob.size());
        if (rc != TH_OK) return rc;
        rc = c->lod_tmp.ensure((n_rows * dw + dw * dh) * sizeof(uint16_t) + 64);
        if (rc != TH_OK) return rc;
        auto axis_dev = [](unsigned char *base, size_t n_out, uint32_t taps) {
            LodAxis a;
            a.start = reinterpret_cast<const int32_t *>(base);
            a.count = reinterpret_cast<const int32_t *>(base + n_out * 4);
            a.wsum = reinterpret_cast<const double *>(base + n_out * 8);
            a.w = reinterpret_cast<const double *>(base + n_out * 16);
            a.n_out = (uint32_t)n_out;
            a.max_taps = taps;
            return a;
        };
        unsigned char *dtab = static_cast<unsigned char *>(c->lod_tabs.dptr);
        uint16_t *d_tmp = static_cast<uint16_t *>(c->lod_tmp.dptr);
        uint16_t *d_lod = d_tmp + ((n_rows * dw + 3) / 4) * 4;
        TH_HIP(launch_lod_hpass(d_img, pitch, (uint32_t)y_lo, (uint32_t)n_rows, axis_dev(dtab, dw, ax.max_taps), d_tmp,
                                (uint32_t)dw, c->stream));
        TH_HIP(launch_lod_vpass(d_tmp, (uint32_t)dw, (uint32_t)y_lo, axis_dev(dtab + bx, dh, ay.max_taps), (uint32_t)dw, d_lod,
                                (uint32_t)dw, c->stream));
        th_raster_desc d{d_lod, (uint8_t *)c->tile_out.dptr, (uint32_t)dw, (uint32_t)dh, 0, 0, (uint32_t)dw, (uint32_t)dh,
                         (uint32_t)dw, 0};
        rc = th_raster_tiles_dev(c, &d, 1, (const uint8_t *)c->colormap.dptr, n_colors);
        if (rc != TH_OK) return rc;
    }
    TH_HIP(hipMemcpyAsync(out + 40, c->tile_out.dptr, g.width * g.height * 4, hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ waveform
TH_API int th_waveform_tiles_dev(th_ctx *c, const th_wave_desc *descs, size_t n) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (n == 0) return TH_OK;
    TH_REQUIRE(descs, "descs is NULL");
    std::vector<WaveJob> jobs(n);
    std::vector<uint32_t> start(n + 1);
    uint64_t blocks = 0;
    for (size_t i = 0; i < n; i++) {
        const th_wave_desc &d = descs[i];
        TH_REQUIRE(d.bin_count <= TH_WAVEFORM_TILE_BINS, "desc %zu: bin_count > 1024", i);
        TH_REQUIRE(d.level < 40, "desc %zu: level %u too large", i, d.level);
        TH_REQUIRE(d.bin_count == 0 || (d.wav && d.bins), "desc %zu: NULL device pointer", i);
        if (d.bin_count) {
            const uint64_t spb = 1ull << d.level;
            TH_REQUIRE(d.start < d.n_samples && d.start + (uint64_t)(d.bin_count - 1) * spb < d.n_samples,
                       "desc %zu: bins run past the end of the channel", i);
        }
        jobs[i] = WaveJob{d.wav, d.bins, d.n_samples, d.start, d.level, d.bin_count};
        start[i] = (uint32_t)blocks;
        blocks += waveform_blocks_for(d.level, d.bin_count);
        TH_REQUIRE(blocks < (1ull << 31), "batch too large for one launch");
    }
    start[n] = (uint32_t)blocks;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    int rc = c->wave_jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(WaveJob));
    if (rc != TH_OK) return rc;
    rc = c->wave_start.upload(c->stream, start.data(), start.size() * sizeof(uint32_t));
    if (rc != TH_OK) return rc;
    TH_HIP(launch_waveform((const WaveJob *)c->wave_jobs.dptr, (const uint32_t *)c->wave_start.dptr, (uint32_t)n,
                           (uint32_t)blocks, c->stream));
    return TH_OK;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ channel statistics
TH_API int th_channel_stats_dev(th_ctx *c, const th_stats_desc *descs, size_t n, float *out_sum_squares,
                                float *out_abs_max) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (n == 0) return TH_OK;
    TH_REQUIRE(descs && out_sum_squares && out_abs_max, "NULL argument");
    TH_REQUIRE(n <= 65535, "at most 65535 channels per call");
    std::vector<StatsJob> jobs(n);
    uint64_t max_samples = 0;
    for (size_t i = 0; i < n; i++) {
        TH_REQUIRE(descs[i].n_samples == 0 || descs[i].wav, "desc %zu: NULL device pointer", i);
        TH_REQUIRE(descs[i].n_samples < (1ull << 40), "desc %zu: too many samples", i);
        jobs[i] = StatsJob{descs[i].wav, descs[i].n_samples, (reinterpret_cast<uintptr_t>(descs[i].wav) & 15u) == 0, 0};
        max_samples = std::max<uint64_t>(max_samples, descs[i].n_samples);
    }
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    int rc = c->pyr_sums.ensure(n * (sizeof(double) + sizeof(uint32_t)));
    if (rc != TH_OK) return rc;
    double *d_sum = reinterpret_cast<double *>(c->pyr_sums.dptr);
    uint32_t *d_pk = reinterpret_cast<uint32_t *>(d_sum + n);
    TH_HIP(hipMemsetAsync(d_sum, 0, n * (sizeof(double) + sizeof(uint32_t)), c->stream));  // sum = 0, peak = +0.0 (:841,876)
    rc = c->pyr_jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(StatsJob));
    if (rc != TH_OK) return rc;
    TH_HIP(launch_channel_stats((const StatsJob *)c->pyr_jobs.dptr, (uint32_t)n, max_samples, d_sum, d_pk, c->stream));
    std::vector<double> hs(n);
    std::vector<uint32_t> hp(n);
    TH_HIP(hipMemcpyAsync(hs.data(), d_sum, n * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipMemcpyAsync(hp.data(), d_pk, n * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; i++) {
        out_sum_squares[i] = (float)hs[i];
        std::memcpy(&out_abs_max[i], &hp[i], sizeof(float));
    }
    return TH_OK;
    TH_CATCH
}

// ------------------------------------------------------------------------------------------ waveform pyramid
TH_API size_t th_waveform_pyramid_bins(uint64_t n_samples, uint32_t level) { return (size_t)pyramid_bins(n_samples, level); }
TH_API size_t th_waveform_pyramid_offset(uint64_t n_samples, uint32_t level) {
    return (size_t)pyramid_offset(n_samples, level < PYR_MAX_LEVELS ? level : PYR_MAX_LEVELS);
}

TH_API int th_waveform_pyramid_dev(th_ctx *c, const th_pyramid_desc *descs, size_t n) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (n == 0) return TH_OK;
    TH_REQUIRE(descs, "descs is NULL");
    TH_REQUIRE(n <= 65535, "at most 65535 channels per call");
    std::vector<PyrJob> jobs(n);
    uint64_t max_samples = 0, sums_total = 0;
    uint32_t max_levels = 0;
    for (size_t i = 0; i < n; i++) {
        const th_pyramid_desc &d = descs[i];
        TH_REQUIRE(d.n_levels <= PYR_MAX_LEVELS, "desc %zu: more than %u levels", i, PYR_MAX_LEVELS);
        TH_REQUIRE(d.n_samples == 0 || d.n_levels == 0 || (d.wav && d.out), "desc %zu: NULL device pointer", i);
        TH_REQUIRE(d.n_samples < (1ull << 40), "desc %zu: too many samples", i);
        PyrJob &j = jobs[i];
        j = PyrJob{};
        TH_REQUIRE(d.first_level <= 2, "desc %zu: first_level must be 0, 1 or 2", i);
        j.wav = d.wav;
        // first_level = 1: the kernels keep addressing level L at out + level_off[L]; shifting `out` back by the extent of the skipped levels
        // (a multiple of 128 bytes) puts level first_level at the start of the caller's buffer; the levels below are never touched
        j.out = d.first_level ? reinterpret_cast<float *>(reinterpret_cast<uintptr_t>(d.out) - pyramid_offset(d.n_samples, d.first_level) * sizeof(float)) : d.out;
        j.n_samples = (d.n_levels > d.first_level) ? d.n_samples : 0;
        j.n_levels = d.n_levels;
        j.aligned16 = ((reinterpret_cast<uintptr_t>(d.wav) & 15u) == 0 ? 1u : 0u) | (d.first_level << 1);
        for (uint32_t l = 0; l < PYR_MAX_LEVELS; l++) j.level_off[l] = pyramid_offset(d.n_samples, l);
        j.sums_half = pyramid_bins(d.n_samples, 12);
        sums_total += 2 * j.sums_half;
        max_samples = std::max<uint64_t>(max_samples, j.n_samples);
        max_levels = std::max(max_levels, d.n_levels);
    }
    if (!max_samples || !max_levels) return TH_OK;
    std::lock_guard<std::recursive_mutex> lk(c->mu);
    TH_HIP(hipSetDevice(c->device));
    int rc = c->pyr_sums.ensure(std::max<uint64_t>(sums_total, 1) * sizeof(float));
    if (rc != TH_OK) return rc;
    uint64_t so = 0;
    for (PyrJob &j : jobs) {
        j.sums = reinterpret_cast<float *>(c->pyr_sums.dptr) + so;
        so += 2 * j.sums_half;
    }
    rc = c->pyr_jobs.upload(c->stream, jobs.data(), jobs.size() * sizeof(PyrJob));
    if (rc != TH_OK) return rc;
    const PyrJob *dj = (const PyrJob *)c->pyr_jobs.dptr;
    TH_HIP(launch_pyramid_base(dj, (uint32_t)n, max_samples, c->stream));
    for (uint32_t l = 13; l < max_levels; l++) TH_HIP(launch_pyramid_up(dj, (uint32_t)n, max_samples, l, (l - 13) & 1u, c->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_encode_waveform_tile_dev(th_ctx *c, const float *d_wav, size_t n_samples, uint64_t revision,
                                       uint32_t level, uint32_t tile_index, uint8_t *out, size_t cap,
                                       size_t *out_len) {
    TH_TRY
    TH_REQUIRE(c && out && out_len, "NULL argument");
    size_t start, bins, spb;
    waveform_tile_geometry(n_samples, level, tile_index, &start, &bins, &spb);
    const size_t need = 24 + bins * 12;
    *out_len = need;
    if (cap < need) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", need);
    put_u64(out, revision);
    put_u32(out + 8, (uint32_t)bins);
    put_u32(out + 12, spb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)spb);
    put_u32(out + 16, tile_index);
    put_u32(out + 20, 0);
    if (bins == 0) return TH_OK;
    TH_REQUIRE(d_wav, "d_wav is NULL");
    std::lock_guard<std::recursive_mutex> lk(c->mu);  // tile_out scratch: whole request
    TH_HIP(hipSetDevice(c->device));
    int rc = c->tile_out.ensure(bins * 12);
    if (rc != TH_OK) return rc;
    // levels whose bins are longer than the channel all give one bin; keep the shift in range
    const uint32_t level_eff = level < 39 ? level : 39;
    th_wave_desc d{d_wav, (float *)c->tile_out.dptr, n_samples, start, level_eff, (uint32_t)bins};
    rc = th_waveform_tiles_dev(c, &d, 1);
    if (rc != TH_OK) return rc;
    TH_HIP(hipMemcpyAsync(out + 24, c->tile_out.dptr, bins * 12, hipMemcpyDeviceToHost, c->stream));
    TH_HIP(hipStreamSynchronize(c->stream));
    return TH_OK;
    TH_CATCH
}
