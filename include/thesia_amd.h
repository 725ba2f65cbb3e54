/*
 * thesia_amd.h — C ABI of the MI355X-native spectrogram / waveform compute path for thesia.
 *
 * This is the drop-in boundary: a plain `extern "C"` surface (opaque handles, plain pointers
 * and sizes, int status codes, no C++/torch types) that replaces the five pure Rust functions
 * at the reference's internal seam and the TrackManager orchestration that calls them.
 * All file:line citations are relative to the reference checkout (Sytronik/thesia).
 *
 *   reference seam                                         → entry point(s) here
 *   ------------------------------------------------------------------------------------------
 *   SpecSetting::calc_framing_params  spectrogram.rs:56-98  → th_calc_framing_params
 *   calc_normalized_win               windows.rs:12-38      → th_calc_normalized_win
 *   calc_mel_fb / _default            src-common/lib.rs:46-103 → th_calc_mel_fb, th_mel_default_n_mel
 *   FreqScale::hz_range_to_idx        src-common/lib.rs:144-159 → th_hz_range_to_idx
 *   SpectrogramAnalyzer prepare/retain spectrogram.rs:101-185 → th_plan_create / th_plan_destroy
 *   SpectrogramAnalyzer::calc_spec    spectrogram.rs:187-212 → th_calc_spec_batch_dev, th_calc_spec_host
 *     (perform_stft stft.rs:16-149, norm :200, mel dot :207, dB decibel.rs:170-214)
 *   find_min_max + range clamp        simd.rs:14-36, core/mod.rs:169-180 → fused into th_calc_spec_*
 *                                                             (per-channel min/max), th_global_db_range
 *   convert_spectrogram_to_img        visualize/drawing.rs:4-33 → th_spec_to_img_dev
 *   encode_spectrogram_tile           render_tiles.rs:281-393 → th_encode_spectrogram_tile_dev, th_raster_tiles_dev
 *   encode_waveform_tile              render_tiles.rs:232-279 → th_encode_waveform_tile_dev, th_waveform_tiles_dev
 *   TrackManager (update_specs, update_spec_imgs, ...) core/mod.rs:33-230 → th_tm_*
 *   tile commands                     src-tauri/src/lib.rs:342-389 → th_tm_get_waveform_tile, th_tm_get_spectrogram_tile
 *
 * Conventions
 *   - Every function returns th_status (0 = ok, <0 = error) and never throws or aborts;
 *     th_last_error() returns a thread-local message for the last failure on this thread.
 *   - The caller owns every host buffer it passes.  Inputs are borrowed for the call only.
 *     Outputs go to caller buffers with a capacity; the written length is returned.
 *   - The library owns device memory behind opaque handles with explicit destroy.
 *   - "_dev" entries take DEVICE pointers, enqueue on the context's HIP stream and do not
 *     synchronise unless they return data to the host.  All other entries take HOST pointers.
 *   - Compute entries (th_calc_spec_*, th_tm_* mutators) may assume exclusive access, like the
 *     reference's single write-lock worker (interface.rs:12-56).  th_tm_* tile getters run
 *     concurrently with each other (each request has its own HIP stream and pinned staging buffer)
 *     and are excluded only while a th_tm_* mutator runs (reader / writer lock, as the reference's
 *     RwLock<TrackManager>, lib.rs:345,378); the context-level th_encode_*_tile_dev entries share
 *     the context stream and serialise on an internal mutex.
 *   - There is NO CPU fallback: compute entries fail with TH_ERR_NO_DEVICE without a GPU.
 */
#ifndef THESIA_AMD_H
#define THESIA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
#define TH_EXTERN_C extern "C"
#else
#define TH_EXTERN_C
#endif
#define TH_API TH_EXTERN_C __attribute__((visibility("default")))

typedef enum {
    TH_OK = 0,
    TH_ERR_INVALID_ARG = -1,
    TH_ERR_UNSUPPORTED = -2,
    TH_ERR_HIP = -3,
    TH_ERR_NO_DEVICE = -4,
    TH_ERR_OOM = -5,
    TH_ERR_BUFFER_TOO_SMALL = -6,
    TH_ERR_NOT_FOUND = -7,
    TH_ERR_INTERNAL = -8
} th_status;

/* FreqScale (src-common/src/lib.rs:105-109) */
#define TH_FREQ_LINEAR 0
#define TH_FREQ_MEL 1

/* render_tiles.rs:14-16 */
#define TH_WAVEFORM_TILE_BINS 1024
#define TH_SPECTROGRAM_TILE_SIZE 512
#define TH_SPECTROGRAM_TILE_GUTTER 4
#define TH_WAVEFORM_TILE_MAX_BYTES (24 + 1024 * 12)
#define TH_SPECTROGRAM_TILE_MAX_BYTES (40 + 520 * 520 * 4)

typedef struct th_ctx th_ctx;   /* device + stream + scratch */
typedef struct th_plan th_plan; /* one SpectrogramAnalyzer cache entry */
typedef struct th_tm th_tm;     /* TrackManager mirror */

/* ---------------------------------------------------------------- errors / info */
TH_API const char *th_last_error(void);
TH_API int th_version(void);
TH_API int th_device_count(int *count);

/* ---------------------------------------------------------------- host-only helpers (no GPU) */
/* SpecSetting::calc_framing_params — spectrogram.rs:56-98 */
TH_API int th_calc_framing_params(double win_ms, uint32_t t_overlap, uint32_t f_overlap, uint32_t sr,
                                  size_t *hop, size_t *win, size_t *n_fft);
/* frame count of perform_stft — stft.rs:50-97 (T = floor((N + 2*(win/2) - win)/hop) + 1) */
TH_API int th_stft_n_frames(size_t n_samples, size_t win, size_t hop, size_t *n_frames);
/* calc_normalized_win(Hann, win, n_fft) — windows.rs:12-38,68-83; out[win] */
TH_API int th_calc_normalized_win(size_t win, size_t n_fft, float *out);
/* calc_mel_fb::<f32> — src-common/src/lib.rs:46-89; out[(n_fft/2+1) * n_mel], fmax < 0 = None */
TH_API int th_calc_mel_fb(uint32_t sr, size_t n_fft, size_t n_mel, float fmin, float fmax, int do_norm,
                          float *out);
/* n_mel chosen by calc_mel_fb_default — src-common/src/lib.rs:91-103 */
TH_API int th_mel_default_n_mel(uint32_t sr, size_t n_fft, size_t *n_mel);
/* FreqScale::hz_range_to_idx — src-common/src/lib.rs:144-159 */
TH_API int th_hz_range_to_idx(int freq_scale, float hz_min, float hz_max, uint32_t sr, size_t n_freqs_or_mels,
                              size_t *i_start, size_t *i_end);
/* max = min(max,0); min = max(min, max - dB_range) over per-spec (min,max) — core/mod.rs:169-180 */
TH_API int th_global_db_range(const float *mins, const float *maxs, size_t n, float dB_range, float *min_dB,
                              float *max_dB);

/* Multi-GPU partitioning of independent (track, channel) units (core/mod.rs:153-163 fans the same
 * units out over rayon threads): deterministic longest-processing-time assignment of units to
 * `world` ranks by weight (frame count).  owner[i] receives the rank of unit i.  No data-path
 * collective is involved; the only exchange of the path is the 2-float dB-range all-reduce. */
TH_API int th_shard_assign(const uint64_t *weights, size_t n_units, uint32_t world, uint32_t *owner);

typedef struct {
    uint32_t width, height;       /* tile size in LOD pixels (0,0 = empty tile) */
    uint32_t origin_x, origin_y;  /* tile origin in LOD pixels */
    uint64_t lod_width, lod_height;
} th_tile_geom;
/* tile geometry of encode_spectrogram_tile — render_tiles.rs:290-313 */
TH_API int th_spectrogram_tile_geometry(size_t img_width, size_t img_height, uint32_t level_x, uint32_t level_y,
                                        uint32_t tile_x, uint32_t tile_y, th_tile_geom *geom);
/* bin geometry of encode_waveform_tile — render_tiles.rs:233-241 */
TH_API int th_waveform_tile_geometry(size_t n_samples, uint32_t level, uint32_t tile_index, size_t *start,
                                     size_t *bin_count, size_t *samples_per_bin);

/* ---------------------------------------------------------------- device context */
/* stream: a hipStream_t (e.g. torch.cuda.current_stream().cuda_stream) or NULL to create one. */
TH_API int th_ctx_create(int device, void *hip_stream, th_ctx **out);
/* use_given_stream != 0: run on `hip_stream` exactly as given, NULL meaning the legacy default stream
 * (what torch.cuda.current_stream() is unless a side stream is active); 0: same as th_ctx_create. */
TH_API int th_ctx_create_ex(int device, void *hip_stream, int use_given_stream, th_ctx **out);
TH_API int th_ctx_destroy(th_ctx *ctx);
TH_API int th_ctx_synchronize(th_ctx *ctx);
/* HIP-graph capture of a sequence of this library's stream-ordered calls on the context's stream (a launch-bound
 * sequence like the 8 small kernels of a single-track update: STFT -> range -> quantise -> raster).  Run the sequence
 * once normally first: descriptor tables and scratch buffers are uploaded / sized on first use, which needs stream
 * synchronisation and is refused while capturing (the call then fails with TH_ERR_HIP and the capture must still be
 * ended).  The graph replays the same launches on the same device pointers; it stays valid while the plans, contexts
 * and buffers it touched are alive and unchanged.  Capture is per thread (hipStreamCaptureModeThreadLocal). */
typedef struct th_graph th_graph;
TH_API int th_ctx_capture_begin(th_ctx *ctx);
TH_API int th_ctx_capture_end(th_ctx *ctx, th_graph **out); /* *out = NULL and an error if the capture was invalidated */
TH_API int th_graph_launch(th_graph *graph);                /* stream-ordered on the context's stream, no sync */
TH_API int th_graph_destroy(th_graph *graph);
/* device memory helpers for callers without their own allocator (tests, C hosts) */
TH_API int th_dev_alloc(th_ctx *ctx, size_t bytes, void **dptr);
TH_API int th_dev_free(th_ctx *ctx, void *dptr);
TH_API int th_dev_upload(th_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
TH_API int th_dev_download(th_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* HIP-event timing on the context's stream (bench.py roofline leg) */
/* stream-ordered device-to-device copy with the library's own 16-byte-per-lane streaming kernel (pointers and size
 * multiples of 16): the copy-bandwidth yardstick bench.py reports beside the roofline fractions */
TH_API int th_dev_copy(th_ctx *ctx, void *d_dst, const void *d_src, size_t bytes);
TH_API int th_timer_start(th_ctx *ctx);
TH_API int th_timer_stop_ms(th_ctx *ctx, float *ms);

/* ---------------------------------------------------------------- SpectrogramAnalyzer plan */
/* largest transform a plan takes: 2^20 samples = 5.5 s at 192 kHz */
#define TH_MAX_N_FFT (1u << 20)
/* Device-resident window / twiddles / mel filterbank for one (sr, win, hop, n_fft, scale, n_mel)
 * key; mirrors prepare()/retain() (spectrogram.rs:116-185).  n_mel = 0 with TH_FREQ_MEL selects
 * calc_mel_fb_default's count.  n_fft: every EVEN value in [2, TH_MAX_N_FFT], win <= n_fft (else TH_ERR_UNSUPPORTED; mel
 * plans whose dense filterbank would exceed 1 GiB are refused the same way): every next_pow2(win) * f_overlap of
 * spectrogram.rs:66-72 up to the limit — powers of two run on the wave / block kernels, 2^a * odd with odd <= 63
 * (f_overlap 3, 5, 6, ...) on the generic kernel with the odd factor as one more pass, larger odd factors (f_overlap 67,
 * 71, 130, ...; round 6) as a chirp-z convolution in double precision.  No UI control offers the last two kinds. */
TH_API int th_plan_create(th_ctx *ctx, uint32_t sr, size_t win, size_t hop, size_t n_fft, int freq_scale,
                          size_t n_mel, th_plan **out);
TH_API int th_plan_destroy(th_plan *plan);
/* n_freq = n_fft/2+1; height = n_freq (linear) or n_mel (mel) = columns of the spec */
TH_API int th_plan_dims(const th_plan *plan, size_t *n_freq, size_t *height);
/* find_min_max over every resident spec (core/mod.rs:169-178) without leaving the device: reduces the n_chan
 * (min, max) pairs a th_calc_spec_batch_dev left in d_minmax to d_out = [min, -max] (2 floats, DEVICE), the form
 * in which ONE element-wise MIN all-reduce merges the ranks of a multi-GPU job.  n_chan = 0 gives [+inf, +inf]. */
TH_API int th_minmax_reduce_dev(th_ctx *ctx, const float *d_minmax, size_t n_chan, float *d_out);
/* update_spec_imgs' range clamp (core/mod.rs:179-180) on the device: d_min_negmax = [min, -max] (DEVICE, as
 * th_minmax_reduce_dev and a MIN all-reduce leave it) -> d_range = [min_dB, max_dB] (DEVICE) with
 * max_dB = min(max, 0), min_dB = max(min, max_dB - dB_range). */
TH_API int th_global_db_range_dev(th_ctx *ctx, const float *d_min_negmax, float dB_range, float *d_range);
/* Both of the above in one launch, for a single GPU (no all-reduce in between): d_range = [min_dB, max_dB];
 * d_min_negmax (may be NULL) additionally receives [min, -max]. */
TH_API int th_minmax_reduce_range_dev(th_ctx *ctx, const float *d_minmax, size_t n_chan, float dB_range,
                                      float *d_min_negmax, float *d_range);
/* name of the kernel th_calc_spec_batch_dev will launch for this plan (for profiles / tests) */
TH_API const char *th_plan_kernel_name(const th_plan *plan);

/* Recommended row pitches (elements) for device-resident specs / images: rows padded to a multiple of
 * 128 bytes so that every row starts on a cache line.  The reference layout (dense rows) is what the
 * copy-out accessors return; the pitch is an HBM-layout choice of this library.
 * Rows laid out at exactly these pitches OWN their padding (elements [row_elems, pitch) of every row): the kernels
 * may fill it with zeros so that the last 128-byte line of a row is written whole (a partially written line costs
 * HBM a read-modify-write: 3.9 -> 5.4 TB/s for 1025-float rows, scripts/ubench/row_stores.hip).  Any other pitch
 * (dense rows, or rows embedded in a wider caller-owned array) is never written outside [0, row_elems). */
TH_API size_t th_pitch_f32(size_t row_elems);
TH_API size_t th_pitch_u16(size_t row_elems);

/* ---------------------------------------------------------------- calc_spec (layer A, device pointers) */
typedef struct {
    const float *wav;    /* DEVICE: n_samples f32, one channel */
    float *spec;         /* DEVICE: n_frames rows of f32 dB, frame-major like the reference's Array2 (T x H) */
    uint64_t n_samples;
    uint64_t n_frames;   /* must equal th_stft_n_frames(n_samples, win, hop) */
    uint64_t spec_pitch; /* floats per row, >= height; 0 = dense (height).  Pitches that are multiples of
                            32 floats (128 B) keep every row cache-line aligned (th_pitch_f32). */
} th_chan_desc;

/* Batched calc_spec over n_chan independent channels (core/mod.rs:153-163 → spectrogram.rs:187-212):
 * reflect-centred framing → Hann/n_fft → real FFT → |X| → [mel] → 20*log10.
 * d_minmax (DEVICE, 2*n_chan f32, may be NULL) receives per-channel (min, max) of the dB values
 * (find_min_max, simd.rs:14-36), fused into the same launch.  Stream-ordered, no sync. */
TH_API int th_calc_spec_batch_dev(th_plan *plan, const th_chan_desc *chans, size_t n_chan, float *d_minmax);
/* The same, and the global dB range of these channels — max_dB = min(max, 0), min_dB = max(min, max_dB - dB_range),
 * core/mod.rs:169-180 — into d_range (DEVICE, 2 floats) in the same call: for a single GPU whose batch is the whole
 * project (one launch fewer than th_calc_spec_batch_dev + th_minmax_reduce_range_dev when the batch is one channel). */
TH_API int th_calc_spec_batch_ranged_dev(th_plan *plan, const th_chan_desc *chans, size_t n_chan, float *d_minmax,
                                         float dB_range, float *d_range);

/* Single-channel convenience with HOST buffers (upload, compute, download; synchronous).
 * out_spec[n_frames * height]; out_min/out_max may be NULL. */
TH_API int th_calc_spec_host(th_plan *plan, const float *wav, size_t n_samples, float *out_spec,
                             size_t out_capacity_floats, size_t *n_frames, float *out_min, float *out_max);

/* ---------------------------------------------------------------- f32 dB spec → u16 grey image */
/* convert_spectrogram_to_img — visualize/drawing.rs:4-33.
 * d_spec: n_frames x height f32; d_img: (i_end - i_start) x n_frames u16 (row 0 = lowest frequency).
 * colormap_len = 0 means None. */
TH_API int th_spec_to_img_dev(th_ctx *ctx, const float *d_spec, size_t n_frames, size_t height, size_t i_start,
                              size_t i_end, float min_dB, float max_dB, uint32_t colormap_len, uint16_t *d_img);

typedef struct {
    const float *spec; /* DEVICE */
    uint16_t *img;     /* DEVICE */
    uint64_t n_frames, height, i_start, i_end;
    uint64_t spec_pitch; /* floats per spec row, 0 = dense (height) */
    uint64_t img_pitch;  /* u16 per image row, 0 = dense (n_frames); multiples of 64 recommended (th_pitch_u16).
                          * NOTE: when img_pitch is a multiple of 64 and img_pitch - n_frames < 64 — i.e. exactly the
                          * library's own padded pitch th_pitch_u16(n_frames) — the kernel owns the row padding and
                          * writes zeros into columns [n_frames, img_pitch) (a whole 128-byte line per store instead of a
                          * read-modify-write).  An image that is a sub-rectangle of a wider surface must therefore use a
                          * pitch with at least 64 columns to the right of it, or a pitch that is not a multiple of 64;
                          * with any such pitch nothing outside [0, n_frames) of a row is written. */
} th_img_desc;
/* batched form: one launch for many channels sharing (min_dB, max_dB, colormap_len) — core/mod.rs:204-227 */
TH_API int th_spec_to_img_batch_dev(th_ctx *ctx, const th_img_desc *descs, size_t n, float min_dB, float max_dB,
                                    uint32_t colormap_len);
/* same with the range left on the device by th_global_db_range_dev (no host round trip between the STFT stage
 * and the quantiser); an all -inf range zero-fills the images as drawing.rs:16-18 */
TH_API int th_spec_to_img_batch_dev_ranged(th_ctx *ctx, const th_img_desc *descs, size_t n, const float *d_range,
                                           uint32_t colormap_len);

/* Quantise + level-0 raster in one pass over the f32 spec (round 4): what th_spec_to_img_batch_dev[_ranged] followed by
 * th_raster_tiles_dev over EVERY level-0 tile of every image produces — the same u16 images, the same RGBA tiles, bit for
 * bit — moving 10 bytes per pixel instead of 6 + 6 (drawing.rs:4-33 + render_tiles.rs:290-351).
 * tiles: HOST array of n_tiles_x * n_tiles_y DEVICE pointers, tile (tx, ty) of the level-0 grid of the (i_end - i_start) x
 * n_frames image (th_spectrogram_tile_geometry with levels 0) at [tx * n_tiles_y + ty]: a dense width x height RGBA8 array,
 * top row = highest frequency, 4-byte aligned (16-byte aligned bases are faster); NULL entries are skipped.
 * d_range != NULL: [min_dB, max_dB] on the DEVICE (th_global_db_range_dev), else the host values.  d_colormap: DEVICE RGBA8
 * LUT of n_colors entries (1 .. 65536); the quantiser's colormap_len is n_colors. */
typedef struct {
    th_img_desc img;
    uint8_t *const *tiles; /* HOST array of DEVICE pointers */
    uint32_t n_tiles_x, n_tiles_y; /* ceil(n_frames / 512), ceil((i_end - i_start) / 512) */
} th_img_tiles_desc;
TH_API int th_spec_to_img_raster_batch_dev(th_ctx *ctx, const th_img_tiles_desc *descs, size_t n, float min_dB, float max_dB,
                                           const float *d_range, const uint8_t *d_colormap, uint32_t n_colors);

/* ---------------------------------------------------------------- tiles */
/* encode_spectrogram_tile — render_tiles.rs:281-352.  d_img: img_height x img_width u16 (DEVICE).
 * colormap: HOST RGBA8 bytes.  Writes the 40-byte LE header + RGBA (top row = highest frequency)
 * to the HOST buffer `out`.  Level (0,0) is an exact crop copy; level > 0 uses a separable
 * Lanczos3 resample in the arithmetic of Pillow's ImagingResample: bit-identical to Pillow 12.2's 16-bit resize on the
 * committed fixtures, formally unpinned vs fast_image_resize itself. */
TH_API int th_encode_spectrogram_tile_dev(th_ctx *ctx, const uint16_t *d_img, size_t img_height, size_t img_width,
                                          size_t img_pitch /* u16 per row, 0 = dense */,
                                          const uint8_t *colormap_rgba, size_t colormap_bytes, uint64_t revision,
                                          uint32_t level_x, uint32_t level_y, uint32_t tile_x, uint32_t tile_y,
                                          uint8_t *out, size_t out_capacity, size_t *out_len);

typedef struct {
    const uint16_t *img;  /* DEVICE: img_height x img_width */
    uint8_t *rgba;        /* DEVICE: height x width x 4, top row = highest frequency; 4-byte aligned (tiles may be
                           * packed back to back: the kernel lays its 16-byte stores on the address grid) */
    uint32_t img_width, img_height;
    uint32_t origin_x, origin_y, width, height; /* level-0 tile rectangle (th_spectrogram_tile_geometry) */
    uint32_t img_pitch;                         /* u16 per image row, 0 = dense (img_width) */
    uint32_t reserved;
} th_raster_desc;
/* Batched level-0 colormap raster of many tile rectangles in one launch (device → device).
 * d_colormap: DEVICE RGBA8, n_colors entries. */
TH_API int th_raster_tiles_dev(th_ctx *ctx, const th_raster_desc *descs, size_t n, const uint8_t *d_colormap,
                               uint32_t n_colors);

/* encode_waveform_tile — render_tiles.rs:232-279.  d_wav: DEVICE samples.  Writes the 24-byte LE
 * header + bins x (min, max, mean) f32 to the HOST buffer `out`. */
TH_API int th_encode_waveform_tile_dev(th_ctx *ctx, const float *d_wav, size_t n_samples, uint64_t revision,
                                       uint32_t level, uint32_t tile_index, uint8_t *out, size_t out_capacity,
                                       size_t *out_len);

typedef struct {
    const float *wav; /* DEVICE */
    float *bins;      /* DEVICE: bin_count x 3 f32 (min, max, mean) */
    uint64_t n_samples;
    uint64_t start;   /* first sample of the tile */
    uint32_t level;
    uint32_t bin_count;
} th_wave_desc;
/* Batched waveform decimation (many tiles / levels / channels in one launch, device → device). */
TH_API int th_waveform_tiles_dev(th_ctx *ctx, const th_wave_desc *descs, size_t n);

/* Channel statistics upstream of the path: sum of squares and absolute peak per channel in one
 * pass — simd.rs:113-183 sum_squares / abs_max as StatCalculator::calc uses them (dynamics/stats.rs:56-86:
 * mean_squared = Σ_ch sum_squares / n_elem, max_peak = max_ch abs_max).  Results are written to HOST arrays. */
typedef struct {
    const float *wav; /* DEVICE */
    uint64_t n_samples;
} th_stats_desc;
TH_API int th_channel_stats_dev(th_ctx *ctx, const th_stats_desc *descs, size_t n, float *out_sum_squares,
                                float *out_abs_max);

/* Waveform pyramid: every decimation level of a channel from one pass over the audio.  Level L
 * (samples per bin 2^L, exactly the bins encode_waveform_tile emits for that level — render_tiles.rs:232-279)
 * has th_waveform_pyramid_bins(n, L) = ceil(n / 2^L) bins of (min, max, mean) f32 and starts at float offset
 * th_waveform_pyramid_offset(n, L) of `out` (every level starts on a 128-byte boundary: the offsets are multiples of 32
 * floats, with up to 31 unused floats behind a level); tile t of level L is bins [1024 t, 1024 (t + 1)) of that level.
 * `out` must hold th_waveform_pyramid_offset(n, n_levels) floats.
 * first_level = F in {1, 2}: levels below F are NOT materialised — level 0 is (x, x, x) per sample, half of all the
 * pyramid's bytes, level 1 a quarter, and a tile of either is no larger as samples (4 / 8 KB) than as bins (12 KB): the
 * TrackManager serves them from the resident audio — and the layout starts at level F: level L >= F at float offset
 * th_waveform_pyramid_offset(n, L) - th_waveform_pyramid_offset(n, F); `out` holds
 * th_waveform_pyramid_offset(n, n_levels) - th_waveform_pyramid_offset(n, F) floats. */
typedef struct {
    const float *wav; /* DEVICE */
    float *out;       /* DEVICE */
    uint64_t n_samples;
    uint32_t n_levels; /* levels 0 .. n_levels-1, at most 40 */
    uint32_t first_level; /* 0 (all levels), 1 or 2 (levels first_level .. n_levels-1); > 2: TH_ERR_INVALID_ARG.  ABI note:
                           * this field was `reserved` (unvalidated) before round 3 — callers must zero it */
} th_pyramid_desc;
TH_API size_t th_waveform_pyramid_bins(uint64_t n_samples, uint32_t level);
TH_API size_t th_waveform_pyramid_offset(uint64_t n_samples, uint32_t level);
TH_API int th_waveform_pyramid_dev(th_ctx *ctx, const th_pyramid_desc *descs, size_t n);

/* ---------------------------------------------------------------- waveform-tile cache (host only, no GPU needed) */
/* Mirror of RenderTileCache — src-tauri/src/core/render_tiles.rs:51-230: a byte-budgeted LRU of encoded
 * waveform tiles keyed by (id, ch, waveform_revision, level, tile_index) plus the waveform / spectrogram
 * revision counters.  th_tm_* owns one; hosts that keep thesia's own lib.rs:342-367 flow can use it directly. */
typedef struct th_tile_cache th_tile_cache;
/* budget_bytes = 0 selects DEFAULT_WAVEFORM_CACHE_BUDGET_BYTES (32 MiB, render_tiles.rs:17) */
TH_API int th_tile_cache_create(size_t budget_bytes, th_tile_cache **out);
TH_API int th_tile_cache_destroy(th_tile_cache *cache);
/* cached_waveform_tile (:124-144): *revision = current waveform revision; *hit = 1 and the bytes on a hit
 * (which also makes the entry most recently used) */
TH_API int th_tile_cache_lookup(th_tile_cache *cache, size_t id, uint32_t ch, uint32_t level, uint32_t tile_index,
                                uint64_t *revision, uint8_t *out, size_t out_capacity, size_t *out_len, int *hit);
/* store_waveform_tile (:146-169): ignored when `revision` is no longer current; evicts least recently used
 * entries until the budget holds (:205-218) */
TH_API int th_tile_cache_store(th_tile_cache *cache, size_t id, uint32_t ch, uint64_t revision, uint32_t level,
                               uint32_t tile_index, const uint8_t *bytes, size_t len);
/* invalidate_waveform (bumps the revision, drops every tile) / invalidate_spectrogram (:87-99) */
TH_API int th_tile_cache_invalidate(th_tile_cache *cache, int waveform, int spectrogram);
TH_API int th_tile_cache_set_budget(th_tile_cache *cache, size_t budget_bytes);
/* any out pointer may be NULL */
TH_API int th_tile_cache_stats(const th_tile_cache *cache, size_t *entries, size_t *bytes, size_t *budget_bytes,
                               uint64_t *waveform_revision, uint64_t *spectrogram_revision, uint64_t *hits,
                               uint64_t *misses);

/* ---------------------------------------------------------------- TrackManager mirror (layer B, host buffers) */
/* Mirrors core/mod.rs:33-230 with decoded audio handed over as planar host f32 (the output of
 * the reference's decode step, audio.rs:65-78).  Audio, f32 dB specs and u16 images stay
 * resident in HBM, so set_dB_range / set_colormap re-quantise without redoing the STFT. */
TH_API int th_tm_create(th_ctx *ctx, th_tm **out);
TH_API int th_tm_destroy(th_tm *tm);
/* init(colormap_rgba) — lib.rs:51-98, render_tiles.rs:80-85; sets colormap_length = bytes/4 */
TH_API int th_tm_set_colormap(th_tm *tm, const uint8_t *rgba, size_t bytes);
/* TrackManager::set_setting — core/mod.rs:107-115 (recomputes every resident track).  Transactional: when the new
 * setting cannot be planned (this library takes every n_fft = next_pow2(win) * f_overlap up to TH_MAX_N_FFT = 2^20 — any
 * f_overlap since round 6; beyond the limit, or with a mel filterbank above 1 GiB, TH_ERR_UNSUPPORTED) or memory runs out, the call
 * fails and the manager — settings, plans,
 * specs, images, revisions — is exactly as before.  th_tm_add_tracks gives the same guarantee. */
TH_API int th_tm_set_setting(th_tm *tm, double win_ms, uint32_t t_overlap, uint32_t f_overlap, int freq_scale);
/* TrackManager::set_dB_range — core/mod.rs:123-126 */
TH_API int th_tm_set_dB_range(th_tm *tm, float dB_range);
/* TrackList::add_tracks + TrackManager::add_tracks — core/mod.rs:62-71.
 * channels[c] points to n_samples f32 of channel c (planar). */
TH_API int th_tm_add_track(th_tm *tm, size_t id, uint32_t sr, uint32_t n_channels, const float *const *channels,
                           size_t n_samples);
/* batch form: computes all added tracks' specs in one launch per plan */
TH_API int th_tm_add_tracks(th_tm *tm, size_t n_tracks, const size_t *ids, const uint32_t *srs,
                            const uint32_t *n_channels, const float *const *channels_flat,
                            const size_t *n_samples);
TH_API int th_tm_remove_track(th_tm *tm, size_t id);
/* TrackManager::apply_track_list_changes — core/mod.rs:102-105,168-230.
 * updated_ids (may be NULL) receives up to cap ids whose images were re-made. */
TH_API int th_tm_apply_track_list_changes(th_tm *tm, size_t *updated_ids, size_t cap, size_t *n_updated,
                                          uint32_t *max_sr);
TH_API int th_tm_get_db_state(const th_tm *tm, float *min_dB, float *max_dB, uint32_t *max_sr);
/* shapes and copy-out accessors (parity tests, get_audio_render_metadata lib.rs:321-340) */
TH_API int th_tm_spec_shape(const th_tm *tm, size_t id, uint32_t ch, size_t *n_frames, size_t *height);
TH_API int th_tm_img_shape(const th_tm *tm, size_t id, uint32_t ch, size_t *img_height, size_t *img_width);
TH_API int th_tm_copy_spec(th_tm *tm, size_t id, uint32_t ch, float *out, size_t capacity_floats);
TH_API int th_tm_copy_img(th_tm *tm, size_t id, uint32_t ch, uint16_t *out, size_t capacity_px);
TH_API int th_tm_revisions(const th_tm *tm, uint64_t *waveform_revision, uint64_t *spectrogram_revision);
/* get_spectrogram_tile / get_waveform_tile — lib.rs:342-389 */
TH_API int th_tm_get_spectrogram_tile(th_tm *tm, size_t id, uint32_t ch, uint32_t level_x, uint32_t level_y,
                                      uint32_t tile_x, uint32_t tile_y, uint8_t *out, size_t out_capacity,
                                      size_t *out_len);
TH_API int th_tm_get_waveform_tile(th_tm *tm, size_t id, uint32_t ch, uint32_t level, uint32_t tile_index,
                                   uint8_t *out, size_t out_capacity, size_t *out_len);
/* Many spectrogram tiles in one call (the initial paint of a view, a zoom that invalidates every visible tile): ONE raster
 * launch and one transfer for all of them instead of a launch + synchronisation per tile.  Tile i is written at
 * out + offsets[i] as the very bytes th_tm_get_spectrogram_tile returns for it (40-byte header + RGBA); records start on
 * 64-byte boundaries, offsets[n] = bytes used, *out_len = bytes needed (TH_ERR_BUFFER_TOO_SMALL when out_capacity is
 * less; out may then be NULL to query the size).  When `out` is pinned host memory (th_host_alloc, or memory the caller
 * registered with HIP) the kernel writes it directly over PCIe; otherwise the tiles pass through a pinned staging buffer
 * of the manager and are copied.  The reference's consumer asks tile by tile (lib.rs:369-389); a host that wants the
 * batch sends its visible-tile list once instead. */
typedef struct {
    size_t id;
    uint32_t ch, level_x, level_y, tile_x, tile_y;
    uint32_t reserved; /* 0 */
} th_tile_request;
TH_API int th_tm_get_spectrogram_tiles(th_tm *tm, const th_tile_request *reqs, size_t n, uint8_t *out, size_t out_capacity,
                                       size_t *offsets /* n + 1 */, size_t *out_len);
/* pinned, device-visible host memory for tile batches (and for audio handed to th_tm_add_tracks: uploads from pinned
 * memory run at PCIe speed) */
TH_API int th_host_alloc(th_ctx *ctx, size_t bytes, void **ptr);
TH_API int th_host_free(th_ctx *ctx, void *ptr);
/* AudioRenderMetadata — render_tiles.rs:36-49, filled as RenderTileCache::metadata (:101-122) does for
 * get_audio_render_metadata (lib.rs:321-340).  track_sec and is_clipped come from the reference's TrackList
 * (decode / clip guard, upstream of this path) and are passed through unchanged; an absent spectrogram gives
 * width = height = 0 (`unwrap_or_default`, :109). */
typedef struct {
    uint64_t waveform_revision, spectrogram_revision;
    uint32_t sample_rate;
    uint32_t is_clipped;
    uint64_t sample_count;
    double track_sec;
    uint64_t spectrogram_width, spectrogram_height;
    uint64_t waveform_tile_bins;    /* WAVEFORM_TILE_BINS = 1024, :14 */
    uint64_t spectrogram_tile_size; /* SPECTROGRAM_TILE_SIZE = 512, :15 */
} th_render_metadata;
TH_API int th_tm_get_audio_render_metadata(th_tm *tm, size_t id, uint32_t ch, double track_sec, int is_clipped,
                                           th_render_metadata *out);
/* the RenderTileCache in front of get_waveform_tile (lib.rs:350-366): borrowed, owned by tm */
TH_API int th_tm_tile_cache(th_tm *tm, th_tile_cache **out);
/* Spectrogram tiles with level_x or level_y > 0 (resize_spectrogram_tile, render_tiles.rs:354-393).
 * Default (per_request = 0): every channel's image gets a mip pyramid when it is (re)made — the whole image resized
 * to ceil(W / 2^lx) x ceil(H / 2^ly) with the separable Lanczos3 — and a LOD tile is a crop of that level + colour
 * LUT, like a level-0 tile.  per_request = 1: the reference's own flow, the tile's crop box is
 * resampled from the level-0 image on every request (also used for levels the pyramid does not hold: images
 * smaller than 16 px at that level); the pyramids are then not kept at all (and come back with per_request = 0).  The
 * two routes agree on whole tiles, core and gutter — both clip the filter at the image, never at the crop box — up to one
 * u16 step in rare pixels (f64 rounding of the tap centres).  PARITY UNPINNED against fast_image_resize in both modes. */
TH_API int th_tm_set_lod_source(th_tm *tm, int per_request);
/* shape of (and, with out != NULL, a dense copy of) one resident mip level; (0, 0) is the image itself */
TH_API int th_tm_mip_level(th_tm *tm, size_t id, uint32_t ch, uint32_t level_x, uint32_t level_y, uint16_t *out,
                           size_t capacity_px, size_t *width, size_t *height);
/* device memory the manager holds besides audio, specs and images (accounting / leak checks): the Lanczos tap tables of
 * the pyramid passes (one per (axis length, level) some resident image needs; dropped with the last such image) and the
 * mip pyramids.  Any out pointer may be NULL. */
TH_API int th_tm_lod_footprint(th_tm *tm, size_t *n_axis_tables, size_t *axis_table_bytes, size_t *mip_bytes);

/* Test and measurement entry points (kernel selectors for A/B runs, per-launch kernel timing, replacing a resident image
 * with given pixels) are NOT part of this interface: include/thesia_amd_testing.h declares them; a thesia host binds none. */

#endif /* THESIA_AMD_H */
