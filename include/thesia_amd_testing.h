/* thesia_amd_testing.h — TESTING / MEASUREMENT entry points of libthesia_amd.so.
 *
 * Nothing here replaces a reference interface and a thesia host binds none of it: these exports exist for
 * this repository's parity tests (tests/), its benchmark (bench.py) and its A/B scripts (scripts/).  They are exported by
 * the same library so that the apparatus measures exactly the code the product runs; they may change between rounds.
 */
#ifndef THESIA_AMD_TESTING_H
#define THESIA_AMD_TESTING_H

#include "thesia_amd.h"

/* ---------------------------------------------------------------- A/B: which kernel a plan launches */
/* kernel selection: 0 = auto, 1 = force the generic workgroup kernel, 2 = force the wave kernel, 3 = wave kernel with
 * the matrix-core mel kernel instead of the fused mel epilogue (mel plans; same as 2 for linear ones), 4 = wave kernel
 * without the grid-aligned register reuse of hop = 480 / 441-style framings, 5 = that reuse also with the fused mel
 * epilogue (both for A/B measurements: it does not pay there), 6 = n_fft 1024 on the two-frames-per-wave plan instead of the
 * one-frame plan (A/B; other sizes: as 2; a wave count in the tuning form below is validated against that plan's launch
 * shapes: 8, 12 or 16), 7 = as 3 with the matrix-core kernel also where 3 runs the
 * banded-sum kernel over the amplitude rows (n_fft 512 under filters of at most 8 bins: the default mel counts of 8-12 kHz audio),
 * 8 = the fused mel epilogue in its pieces / gather form where banded sums are the default (A/B),
 * 9 = the wave kernel's packed-f32 pipeline (v_pk_fma_f32 butterflies on register pairs) where it is instantiated: n_fft 2048,
 * hop = n_fft / 4, linear dB, default waves (A/B: it measures the same as the scalar pipeline; elsewhere as 2),
 * 11 = the wave kernel with the "sweep" chunk schedule (4-frame chunks dealt out in order through a per-workgroup ticket
 * counter, the next chunk's first frame prefetched) on large batches of that same shape (A/B: a faster memory skeleton, the
 * same launch time; elsewhere as 2; 10 is reserved and behaves as 2),
 * 12 = mel plans at n_fft 4096: the banded sums as the FFT kernel's epilogue with the table read from global memory (L2) instead of
 * the second kernel over amplitude rows, on the launch shapes it is instantiated for (hop 1024 and the 96 / 88.2 kHz defaults,
 * at most 512 mels; A/B: measured slower, profiles/r05_ab_mel4096_fused.txt; elsewhere as 2),
 * 13 = the fused mel epilogue one frame at a time where frame pairs are the default (n_fft 1024 / 2048 banded sums; A/B, bit-identical),
 * 14 = the workgroup-per-frame Stockham kernels (stft_block_kernel / its planar form) where stft_subwave_kernel is the default
 * (n_fft 32768, 65536, 16384 at hops other than n_fft / 4; A/B),
 * 15 = stft_subwave_kernel (R 1024-point wave transforms + a combining pass) wherever it exists, n_fft 8192 .. 65536 (A/B);
 * tuning: (chunk << 16) | (waves << 8) | 2 runs the wave kernel with 4..16 waves per workgroup and `chunk`
 * frames per queue pull */
TH_API int th_plan_set_kernel(th_plan *plan, int which);

/* ---------------------------------------------------------------- measurement: the dominant kernel's launch duration */
/* Measurement hook: with enable != 0 every th_calc_spec_batch_dev records two HIP events on the context's stream
 * around its dominant kernel launch (the wave kernel, or the generic one when that is all there is);
 * recording does not synchronise.  th_plan_kernel_ms_history returns the durations of the most recent launches
 * (oldest first, at most 64 are kept; it waits for them), th_plan_last_kernel_ms the latest one.
 * th_plan_time_kernel also resets the history. */
TH_API int th_plan_time_kernel(th_plan *plan, int enable);
TH_API int th_plan_kernel_ms_history(th_plan *plan, float *out_ms, size_t capacity, size_t *n_out);
TH_API int th_plan_last_kernel_ms(th_plan *plan, float *ms);

/* ---------------------------------------------------------------- parity tests: given pixels under a TrackManager */
/* Replaces the pixels of one resident u16 image (same shape: H x W dense u16, row 0 = lowest frequency) and rebuilds its mip
 * pyramid, as update_spec_imgs does after a re-quantise (core/mod.rs:181-229).  For hosts that quantise elsewhere and for
 * the parity tests, which pin the pyramid to third-party known answers on given images (tests/golden/lod_pillow_cases.npz);
 * the next update_spec_imgs of that channel overwrites it.  Bumps the spectrogram revision. */
TH_API int th_tm_put_img(th_tm *tm, size_t id, uint32_t ch, const uint16_t *img, size_t height, size_t width);

#endif /* THESIA_AMD_TESTING_H */
