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
 * 12 = mel plans whose default is the moment-form epilogue (n_fft 4096 / 8192 / 16384; n_fft 512 .. 2048 where no LDS table form exists): the two kernels (FFT kernel -> amplitude rows -> banded sums / matrix cores) where the moment-form
 * epilogue is the default (hop 1024, the 96 / 88.2 kHz defaults; round 5's route, kept for A/B; elsewhere as 2),
 * 13 = the fused mel epilogue one frame at a time where frame pairs are the default (n_fft 1024 / 2048 banded sums; A/B, bit-identical),
 * 14 = the workgroup-per-frame Stockham kernel where stft_subwave_kernel is the default (n_fft 16384 at hops other than n_fft / 4),
 * 15 = stft_subwave_kernel (R 1024-point wave transforms + a combining pass) at n_fft 16384 also at hop n_fft / 4 (A/B);
 * 10 is reserved and behaves as 2;
 * tuning: (chunk << 16) | (waves << 8) | 2 runs the wave kernel with `waves` waves per workgroup (each size's own shapes: n_fft
 * 1024 / 2048: 12; n_fft 4096: 8 or 7; the multi-frame plans 8, 12, 16) and `chunk` frames per queue pull.
 * Only in libraries built with -DTH_AB_VARIANTS=1 (th_build_ab_variants() == 1; scripts/build_variant.sh), TH_ERR_UNSUPPORTED
 * otherwise — measured, dropped, and no longer part of the product binary (HISTORY.md has the numbers):
 * 9 = the wave kernel's packed-f32 pipeline (v_pk_fma_f32 butterflies on register pairs; n_fft 2048, hop = n_fft / 4, linear dB),
 * 11 = the "sweep" chunk schedule (4-frame chunks dealt out in order through a per-workgroup ticket counter) on large batches of
 * that shape, 14 at n_fft 32768 / 65536 (stft_block_kernel / its planar form), 15 at n_fft 8192, and every other waves-per-workgroup
 * shape of the one-frame wave kernels (4 .. 16) */
TH_API int th_plan_set_kernel(th_plan *plan, int which);

/* 1 when the library carries the A/B variants above (-DTH_AB_VARIANTS=1), 0 for the product build */
TH_API int th_build_ab_variants(void);

/* Mel plans: the moment-form table of the fused epilogue, if the plan has one (n_fft 4096 / 8192 / 16384, and n_fft 512 .. 2048 where no LDS table form exists; n_groups = 0: it does not —
 * not such a plan, or the filterbank's lines leave the reference's f32 weights by more than the builder allows, and the plan keeps
 * the two kernels).  taps = segment taps per frame, max_dev = the largest difference between a line and the table's weight over
 * all bins (units of an unnormalised weight), max_amp = how far a filter's 1 / d enlarges the moments' rounding.  Any may be NULL. */
TH_API int th_plan_mel_moments_info(const th_plan *plan, uint32_t *n_groups, uint32_t *taps, double *max_dev, double *max_amp);

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
