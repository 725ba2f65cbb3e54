// kernels.h — launcher declarations (implemented in kernels_*.hip).  All launchers are
// stream-ordered, never synchronise, and return the launch status.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "../../include/thesia_amd.h"
#include "stft_core.h"

// Measured-and-dropped kernel variants: the packed-f32 pipeline (selector 9), the sweep chunk schedule (11), the
// workgroup-per-frame Stockham kernels of n_fft 32768 / 65536 (14), the subwave plan at n_fft 8192 (15) and launch shapes other
// than each size's defaults (waves-per-workgroup tuning) are compiled only with -DTH_AB_VARIANTS=1 (scripts/build_variant.sh
// <tag> -DTH_AB_VARIANTS=1 ...); the product library carries the kernels a default route can reach.  th_build_ab_variants()
// reports which build this is; th_plan_set_kernel refuses the selectors that are not in it.
#if !defined(TH_AB_VARIANTS)
#define TH_AB_VARIANTS 0
#endif
// The banded mel table of the n_fft 1024 / 2048 wave kernels in its paired layout (mel_fuse.h build_mel_band, stft_wave.h
// mel_banded<true>): host table and device code must agree, so the switch lives here (0: plain layout, for A/B builds; the
// frame-pair epilogue reads the paired layout only and is then off: stft_wave_mel_pair_applies).
#if !defined(TH_MEL_BAND_PAIRED)
#define TH_MEL_BAND_PAIRED 1
#endif

// The same switch for mel_band_rows_kernel (n_fft 4096, two-kernel path): table from api.hip, reads in kernels_mel.hip.
// (2.139 against 2.158 ms at the 96 kHz default, 2.152 against 2.163 at 88.2 kHz: the kernel is bound by reading the amplitude rows)
#if !defined(TH_MEL_ROWS_PAIRED)
#define TH_MEL_ROWS_PAIRED 1
#endif

namespace th {

// ---- kernels_stft.hip
hipError_t launch_minmax_init(float *d_minmax, uint32_t n_chan, hipStream_t s);
// d_out = [min, -max] and / or d_range = [min_dB, max_dB] (core/mod.rs:179-180); either may be NULL
hipError_t launch_minmax_reduce(const float *d_minmax, uint32_t n_chan, float *d_out, float dB_range, float *d_range, hipStream_t s);
size_t stft_generic_lds_bytes(const StftGeom &g);
// n_fft >= 32768: the generic kernel keeps its two frame buffers in global scratch (d_scratch, at least
// stft_generic_scratch_bytes) instead of LDS and runs as a persistent grid; 0 bytes = the LDS variant, no scratch needed
size_t stft_generic_scratch_bytes(const StftGeom &g, uint32_t n_tiles, uint32_t n_cu);
// Bluestein plans (an odd factor of n_fft above 63; stft_bluestein_kernel): convolution length, scratch bytes of a launch, launch.
// d_chirp[nc], d_bhat[M], d_twm[M / 2], d_tws[nc / 2 + 1]: complex doubles (bluestein_tables, host_math.h)
constexpr uint32_t STFT_MAX_DIRECT_ODD = 63;  // the largest odd factor the generic kernel takes as one direct pass
uint32_t stft_bluestein_m(const StftGeom &g);
size_t stft_bluestein_scratch_bytes(const StftGeom &g, uint32_t n_tiles, uint32_t n_cu);
hipError_t launch_stft_bluestein(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan, uint32_t n_tiles,
                                 const float *d_window, const void *d_chirp, const void *d_bhat, const void *d_twm, const void *d_tws,
                                 const float *d_mel_fb, const uint32_t *d_mel_lo, const uint32_t *d_mel_hi, float *d_minmax, hipStream_t s,
                                 void *d_scratch, uint32_t n_cu);
hipError_t launch_stft_generic(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start,
                               uint32_t n_chan, uint32_t n_tiles, const float *d_window, const cf32 *d_tw,
                               const float *d_mel_fb, const uint32_t *d_mel_lo, const uint32_t *d_mel_hi,
                               float *d_minmax, hipStream_t s, void *d_scratch = nullptr, uint32_t n_cu = 256);

bool stft_wave_supported(const StftGeom &g);
// n_fft 8192 / 16384: launch_stft_wave runs the workgroup-per-frame kernel (stft_block.h): interior frames only, the
// boundary frames always go to the generic kernel
bool stft_is_block_plan(const StftGeom &g);
// n_fft 8192 / 16384 mel plans: the filterbank (moment form) in the block kernel's epilogue where that kernel is the plan that runs
bool stft_block_mel_fused_applies(const StftGeom &g, int long_plan);
uint32_t stft_block_mel_max_index(const StftGeom &g);
// the multi-frame kernel (two / four frames per wave) takes this launch: n_fft 512 or 1024, dB output, no grid-aligned mode
bool stft_wave_multi_applies(const StftGeom &g, int out_mode);
uint32_t stft_wave_multi_tail_guard(const StftGeom &g);  // samples an interior span must stay clear of the channel's end
int stft_wave_default_waves(const StftGeom &g);
// d_minmax of launch_stft_wave is a per-CHUNK (min, max) array (2 floats per tile).  launch_wave_post follows every wave
// launch: per channel it folds the chunk pairs of the channel's tiles [t0, t1) into the channel slot (store = the slot
// needs no initialisation: every frame of the channel was in the wave launch) and rewinds the queue (d_queue_head[0],
// zero before the first launch; the kernel counts from 0 behind its statically assigned first round of chunks: a wave's
// first chunk is its own index, TH_SCHED_ADVANCE adds the grid's wave count to what it pulls).
struct WavePostJob {
    uint32_t t0, t1, mm_index, reserved;
};
// d_range (may be NULL; single-job launches with store only): [min_dB, max_dB] of the one channel, clamped by dB_range
hipError_t launch_wave_post(const WavePostJob *d_pj, uint32_t n_pj, const float *d_chunk_mm, float *d_mm_slots, bool store,
                            uint32_t *d_queue_head, float dB_range, float *d_range, hipStream_t s);
// What the wave kernel writes: dB rows of the linear spectrum, linear amplitude rows (first half of the matrix-core
// mel path), or dB rows of the mel spectrum with the filterbank fused into the epilogue (mel_fuse.h tables).
struct WaveOut {
    int mode = 0;                     // 0 dB linear, 1 amplitude, 2 fused mel
    int multi = 0;                    // 1: the multi-frame kernel (n_fft 512 / 1024, stft_wave_multi.h; mode 0 only)
    int packed = 0;                   // 1: the packed-f32 pipeline (stft_pk.h) where it is instantiated (selector 9), else ignored
    int sweep = 0;                    // 1: the sweep chunk schedule (4-frame chunks dealt out in order; n_fft 2048, hop = n_fft / 4, dB rows, 12 waves)
    const uint32_t *mel_tab = nullptr;  // DEVICE: mel_fuse.h word table
    uint32_t mel_words = 0, mel_slots = 0, mel_groups = 0, n_mel = 0;
    // banded sums (mel_slots == 0, build_mel_band): the table's header again, as kernel arguments (scalar registers instead
    // of two LDS reads + v_readfirstlane per group and frame): block offset and taps of group g
    uint32_t band_off[8] = {0, 0, 0, 0, 0, 0, 0, 0}, band_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // banded sums, n_fft 2048: 1 = the frame-pair epilogue (two consecutive frames of a chunk share one pass over the table:
    // stft_wave_kernel<.., OUT = 3>; the host sets it where stft_wave_mel_pair_applies says so)
    uint32_t mel_pair = 0;
    // n_fft 1024 / 2048: 1 = the moment form of the filterbank (mel_tab = build_mel_moments' table, read from global memory; mel_words = 0):
    // mel counts above what an LDS table holds.  (n_fft 4096 .. 16384 mel plans are always this form.)
    uint32_t mel_moment = 0;
    const cf32 *subwave_twc = nullptr;  // DEVICE: stft_subwave_build_twc's table (n_fft 32768 plans)
    int long_plan = 0;  // n_fft 8192 .. 32768: 0 = the default of the size (stft_subwave_default), 1 = stft_block_kernel, 2 = stft_subwave_kernel
};
// ---- kernels_stft_long.hip: n_fft 8192 / 16384 / 32768 as 4 / 8 / 16 wave transforms + one combining pass (interior frames, as the block kernel)
bool stft_subwave_applies(const StftGeom &g);
bool stft_subwave_default(const StftGeom &g);  // the sizes where it is the default (measured: profiles/r05_ab_subwave.txt)
hipError_t launch_stft_subwave(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_chunk_tab, uint32_t n_tiles, const cf32 *d_wtab,
                               const cf32 *d_tw, const cf32 *d_twc, float *d_minmax, bool amp, uint32_t n_cu, hipStream_t s);
size_t stft_subwave_twc_len(const StftGeom &g);  // cf32 entries of the combining pass's per-thread constant table
void stft_subwave_build_twc(const StftGeom &g, const cf32 *h_tw, cf32 *out);  // host: out[stft_subwave_twc_len(g)] from the plan's tw table
// the frame-pair form of the banded mel epilogue exists for this launch (n_fft 2048, hop a multiple of 128 samples or no register
// reuse at all, default waves) and the table's reads stay inside the two amplitude rows a slab holds (reach: MelBandHost::reach)
bool stft_wave_mel_pair_applies(const StftGeom &g, int waves, uint32_t reach);
// pieces the per-wave (r, f) buffer of the fused mel epilogue can hold for this n_fft (0: not supported), and whether
// the launch shape leaves room in LDS for a table of `words`
int stft_wave_phased_mode(const StftGeom &g, int waves);  // 0 no, 1 phased (hop 480), 2 dynamic (e.g. 441)
int stft_wave_mel_phase_mode(const StftGeom &g, int waves);  // the same for a launch with the fused mel epilogue (n_fft 4096: only the modes it is instantiated for)
uint32_t stft_wave_mel_max_pieces(const StftGeom &g);
// the sweep chunk schedule exists for this launch (shape, output mode, wave count; the caller adds: batch large enough)
bool stft_wave_sweep_applies(const StftGeom &g, int waves, int out_mode);
// banded: the banded-sum table (no per-wave (r, f) buffer at n_fft 1024, which only the pieces / gather form uses)
bool stft_wave_mel_fits(const StftGeom &g, int waves, uint32_t words, bool banded = false);
// n_fft 512 (four frames per wave): does the mel_rows table fit the launch's LDS beside the slabs at this wave count?
bool stft_wave_multi_mel_fits(const StftGeom &g, int waves, uint32_t words);
uint32_t stft_wave_multi_amp_pitch();  // floats per amplitude row of that kernel's mel epilogue (a lane of the moment form reads inside its frame's row)
// waves: waves per workgroup (4, 8, 12 or 16; <= 0 selects the default for this n_fft)
// d_tile_start: HERE the per-chunk table, 2 words per chunk: (job, first frame) — not the jobs' first chunks as in
// launch_stft_generic; chunks 0 .. W - 1 are statically assigned (W = waves of the grid), the queue serves the rest
hipError_t launch_stft_wave(const StftGeom &g, const ChanJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_chan,
                            uint32_t n_tiles, const cf32 *d_wtab, const cf32 *d_tw, float *d_minmax,
                            uint32_t *d_queue_head, uint32_t n_cu, int waves, const WaveOut &out, hipStream_t s);

// ---- kernels_mel.hip: mel filterbank contraction on the matrix cores (spectrogram.rs:207)
struct MelJob {
    const float *amp;  // n_frames x amp_pitch linear amplitudes |X| (columns >= n_freq are zero)
    float *spec;       // n_frames x spec_pitch dB mel spectrogram
    uint32_t f_begin, f_end, spec_pitch, mm_index;
};
#if !defined(TH_MEL_MT)
#define TH_MEL_MT 2
#endif
constexpr int MEL_MT = TH_MEL_MT;                         // 16-frame row tiles per wave
constexpr uint32_t MEL_TILE_FRAMES = 64 * MEL_MT;         // 4 waves x MEL_MT x 16 frames per workgroup
hipError_t launch_mel_mfma(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                           uint32_t amp_pitch, const float *d_bt, const uint32_t *d_tile_band, const uint32_t *d_slice_start,
                           uint32_t n_slices, uint32_t zero_block, uint32_t n_mel, float *d_minmax, hipStream_t s);
// short rows (at most MEL_ROWS_NKB * 16 bins: n_fft 512) under narrow filters (at most MEL_ROWS_W bins each, at most
// 64 * MEL_ROWS_MAX_GROUPS mels): banded sums, lane = mel.  d_tab: [n_groups][1 + MEL_ROWS_W][64] words — the first bin of
// mel 64 g + lane, then its weights (float bits, zero past the filter's end and for mels >= n_mel)
constexpr int MEL_ROWS_NKB = 17, MEL_ROWS_W = 8, MEL_ROWS_MAX_GROUPS = 8;
hipError_t launch_mel_rows(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                           uint32_t amp_pitch, const uint32_t *d_tab, uint32_t n_groups, uint32_t n_mel, float *d_minmax,
                           uint32_t n_cu, hipStream_t s);

// long rows (n_fft 4096) under narrow filters: the banded sums of build_mel_band (mel_fuse.h) over the amplitude rows,
// one row per wave at a time through LDS.  hdr: the table's 16 header words (offset and taps per group).
bool mel_band_rows_fits(uint32_t n_freq, uint32_t words);
hipError_t launch_mel_band_rows(const MelJob *d_jobs, const uint32_t *d_tile_start, uint32_t n_jobs, uint32_t n_tiles,
                                uint32_t amp_pitch, uint32_t n_freq, const uint32_t *d_tab, uint32_t words, uint32_t groups,
                                const uint32_t *hdr, uint32_t n_mel, float *d_minmax, uint32_t n_cu, hipStream_t s);

// ---- kernels_image.hip
struct ImgJob {  // device-visible copy of th_img_desc
    const float *spec;
    uint16_t *img;
    uint32_t n_frames, height, i_start, i_end;
    uint32_t spec_pitch, img_pitch;  // elements per row (>= height / n_frames)
    uint32_t first_tile, n_tiles;    // this job's block range in the launch
};
hipError_t launch_spec_to_img(const ImgJob *d_jobs, const uint32_t *d_tile_job, uint32_t n_jobs,
                              uint32_t n_tiles, float min_dB, float max_dB, uint32_t colormap_len, const float *d_range,
                              hipStream_t s);
hipError_t launch_db_range(const float *d_in, float dB_range, float *d_out, hipStream_t s);

// Quantise + level-0 raster in one pass (round 4): one job per image; block b of a job = (tile column tx = local / n_bands,
// band of FUSED_FB image rows = local % n_bands); tiles[tile0 + tx * n_ty + ty] = RGBA array of level-0 tile (tx, ty) or NULL
struct FusedJob {
    const float *spec;
    uint16_t *img;
    uint32_t n_frames, height, i_start, i_end;
    uint32_t spec_pitch, img_pitch;
    uint32_t first_block, n_bands;
    uint32_t n_tx, n_ty;
    uint32_t tile0, reserved;
};
static_assert(sizeof(FusedJob) == 64, "FusedJob must have no implicit padding");
#if !defined(TH_FUSED_FB)
#define TH_FUSED_FB 32
#endif
#if !defined(TH_FUSED_THREADS)
#define TH_FUSED_THREADS 256
#endif
constexpr uint32_t FUSED_FB = TH_FUSED_FB;        // image rows (frequency bins) per block
constexpr uint32_t FUSED_THREADS = TH_FUSED_THREADS;
hipError_t launch_spec_to_img_raster(const FusedJob *d_jobs, const uint32_t *d_block_job, uint32_t n_blocks, uint8_t *const *d_tiles,
                                     float min_dB, float max_dB, const float *d_range, int all_zero, const uint8_t *d_colormap,
                                     uint32_t n_colors, hipStream_t s);

struct RasterJob {  // device-visible copy of th_raster_desc (+ derived fields)
    const uint16_t *img;
    uint8_t *rgba;
    uint32_t img_width, img_height, origin_x, origin_y, width, height;
    uint32_t img_pitch;      // u16 elements per image row (>= img_width)
    uint32_t quads_per_row;  // ceil(width / 4): a thread rasterises 4 horizontally adjacent pixels
    uint32_t inv_qpr;        // floor(2^32 / quads_per_row) + 1: q / quads_per_row == umulhi(q, inv_qpr)
    uint32_t inv_width;      // floor(2^32 / width) + 1 (flat-quad path of widths that are not multiples of 4)
    uint32_t first_block;    // this job's first block in the launch
};
hipError_t launch_raster_level0(const RasterJob *d_jobs, const uint32_t *d_block_job, uint32_t n_jobs,
                                uint32_t n_blocks, const uint8_t *d_colormap, uint32_t n_colors, hipStream_t s);
#if !defined(TH_RASTER_THREADS)
#define TH_RASTER_THREADS 256
#endif
#if !defined(TH_RASTER_QPB)
#define TH_RASTER_QPB 1024
#endif
constexpr uint32_t RASTER_THREADS = TH_RASTER_THREADS;
constexpr uint32_t RASTER_QUADS_PER_BLOCK = TH_RASTER_QPB;  // default: 256 threads x 4 quads of 4 pixels
constexpr uint32_t IMG_TILE_T = 64;   // frames per quantise/transpose tile
constexpr uint32_t IMG_TILE_F = 128;  // frequency rows per tile

// Separable Lanczos3 LOD resample of a crop of a u16 image (encode_spectrogram_tile, LOD > 0,
// render_tiles.rs:354-393).  Per output index of an axis the host tabulates the first source index,
// the tap count, the f64 tap weights and their sum (same order as the CPU restatement).
struct LodAxis {             // device pointers into one uploaded blob
    const int32_t *start;    // [n_out] first source index (already clamped to the image / row window)
    const int32_t *count;    // [n_out] number of taps
    const double *wsum;      // [n_out] sum of the taps
    const double *w;         // [n_out][max_taps]
    uint32_t n_out, max_taps;
    // batched pass only (launch_lod_vpass_batch): rows spanned by the tap windows of R consecutive outputs, maximum over
    // the aligned groups of R = 8, 4, 2 outputs (0: unknown, one output per thread)
    uint32_t span[3] = {0, 0, 0};
};
// horizontal pass: tmp[r][ox] for the source rows y_lo + r, r < n_rows; vertical pass: lod[oy][ox]
// (tmp_pitch / lod_pitch: elements per row of the intermediate and of the result)
hipError_t launch_lod_hpass(const uint16_t *d_img, uint32_t img_pitch, uint32_t y_lo, uint32_t n_rows, LodAxis ax,
                            uint16_t *d_tmp, uint32_t tmp_pitch, hipStream_t s);
hipError_t launch_lod_vpass(const uint16_t *d_tmp, uint32_t tmp_pitch, uint32_t y_lo, LodAxis ay, uint32_t dw,
                            uint16_t *d_lod, uint32_t lod_pitch, hipStream_t s);
// the vertical pass over many images of one shape (whole images: source rows from 0): one launch, grid z = job
struct LodPassJob {
    const uint16_t *src;
    uint16_t *dst;
    uint32_t src_pitch, dst_pitch;
};
hipError_t launch_lod_vpass_batch(const LodPassJob *d_jobs, uint32_t n_jobs, LodAxis ay, uint32_t dw, hipStream_t s);
// dst[x][y] = src[y][x] for w x h images (src_pitch / dst_pitch in elements)
hipError_t launch_transpose_u16_batch(const LodPassJob *d_jobs, uint32_t n_jobs, uint32_t w, uint32_t h, hipStream_t s);
// one tile rectangle of one image -> RGBA, job passed by value (tile requests: nothing to upload)
hipError_t launch_raster_tile(const uint16_t *d_img, uint32_t img_width, uint32_t img_height, uint32_t img_pitch,
                              uint32_t origin_x, uint32_t origin_y, uint32_t width, uint32_t height, uint8_t *d_rgba,
                              const uint8_t *d_colormap, uint32_t n_colors, hipStream_t s);

// streaming 16-byte-per-lane device copy of bytes / 16 * 16 bytes (the bandwidth yardstick of bench.py)
hipError_t launch_copy_f4(const void *d_src, void *d_dst, uint64_t bytes, hipStream_t s);

// ---- kernels_waveform.hip
struct WaveJob {  // device-visible copy of th_wave_desc
    const float *wav;
    float *bins;
    uint64_t n_samples, start;
    uint32_t level, bin_count;
};
hipError_t launch_waveform(const WaveJob *d_jobs, const uint32_t *d_block_start, uint32_t n_jobs,
                           uint32_t n_blocks, hipStream_t s);

// waveform pyramid (all levels of a channel in one pass over the audio)
constexpr uint32_t PYR_MAX_LEVELS = 40;
struct PyrJob {
    const float *wav;
    float *out;          // all levels, level L at float offset level_off[L]
    float *sums;         // scratch: 2 * sums_half floats (bin sums of the level being reduced, ping-pong)
    uint64_t n_samples;
    uint64_t sums_half;
    uint64_t level_off[PYR_MAX_LEVELS];
    uint32_t n_levels;
    uint32_t aligned16;  // bit 0: wav is 16-byte aligned (float4 loads); bits 1-2: th_pyramid_desc.first_level (levels below it are not written)
};
struct StatsJob {
    const float *wav;
    uint64_t n_samples;
    uint32_t aligned16, pad_;
};
hipError_t launch_channel_stats(const StatsJob *d_jobs, uint32_t n_jobs, uint64_t max_samples, double *d_sumsq,
                                uint32_t *d_peak_bits, hipStream_t s);
uint64_t pyramid_bins(uint64_t n, uint32_t level);
uint64_t pyramid_offset(uint64_t n, uint32_t level);
hipError_t launch_pyramid_base(const PyrJob *d_jobs, uint32_t n_jobs, uint64_t max_samples, hipStream_t s);
hipError_t launch_pyramid_up(const PyrJob *d_jobs, uint32_t n_jobs, uint64_t max_samples, uint32_t level, uint32_t parity,
                             hipStream_t s);
uint32_t waveform_blocks_for(uint32_t level, uint32_t bin_count);

}  // namespace th
