"""LOD > 0 tiles (render_tiles.rs:290-313,354-393) — what can be established WITHOUT the fast_image_resize source.

PARITY UNPINNED against fast_image_resize 6.0.0 (crate not vendored).  Two things are pinned down here instead:

1. *Position purity.*  The reference resamples the crop box of one tile, with the filter clipped at the IMAGE
   (fast_image_resize / Pillow semantics: taps outside the crop box are read, taps outside the image are dropped).  A pixel
   of LOD level (lx, ly) at LOD coordinates (X, Y) therefore has one value whichever tile it is requested through — core
   or gutter.  So a pre-built mip level (the whole image as the crop box) must reproduce every per-request tile INCLUDING
   its 4-pixel gutters, up to f64 rounding of the tap centres (<= 1 u16 step in rare pixels).  This is the property the
   TrackManager's mip pyramid relies on (thesia_amd/csrc/track_manager.hip: build_mips).

2. *Sensitivity bound.*  fast_image_resize resamples u16 images in fixed point (i32 coefficients, i64 accumulation, round
   half up).  The oracle restates that structure at two precisions (the largest an i32 coefficient allows, and a crude
   16 bits) and the tests bound how far such a filter can sit from the f64 one: on the u16 plane and on the colour-index
   plane the frontend sees (258-entry colour map: one colour step = 255 u16 steps).  This is a bound on the sensitivity to
   the un-vendored crate's arithmetic, not parity with it.
"""
import numpy as np
import pytest

from oracle import oracle as orc
from tests.synth import synth_track

C = 258  # colour-map entries of the app (src/prototypes/constants/colors.ts:65-165)


def _ci(v):
    return (v.astype(np.int64) * (C - 1) + 32767) // 65535  # render_tiles.rs:345


def _spec_image(seed: int, n: int, win=1920, hop=480, n_fft=2048, mel=True):
    x = synth_track(seed, 48000, n)
    fb = orc.calc_mel_fb_default(48000, n_fft) if mel else None
    spec = orc.calc_spec(x, win, hop, n_fft, mel_fb=fb, fft32=True)
    mn, mx = orc.find_min_max(spec)
    lo, hi = orc.global_db_range([mn], [mx], 100.0)
    return orc.convert_spectrogram_to_img(spec, (0, spec.shape[1]), (lo, hi), C)


@pytest.fixture(scope="module")
def images():
    rng = np.random.default_rng(5)
    return {
        "spectrogram": _spec_image(32, 48000 * 16),                         # 1601 frames x 347 mels: four tile columns
        "noise": rng.integers(0, 65536, (700, 1333)).astype(np.uint16),     # worst case for any resampler; two tile rows
        "steps": np.repeat(np.repeat(rng.integers(0, 2, (40, 70)) * 65535, 9, 0), 17, 1).astype(np.uint16),  # hard edges
    }


LEVELS = [(1, 0), (2, 0), (0, 1), (1, 1), (2, 1), (3, 2)]


@pytest.mark.parametrize("name", ["spectrogram", "noise", "steps"])
def test_whole_image_resize_reproduces_every_tile_including_gutters(images, name):
    img = images[name]
    worst, n_diff, n_px, n_gutter = 0, 0, 0, 0
    for lx, ly in LEVELS:
        whole = orc.resize_whole_image(img, lx, ly)
        for ty in range(-(-whole.shape[0] // 512)):
            for tx in range(-(-whole.shape[1] // 512)):
                t, (ox, oy) = orc.spectrogram_tile_u16(img, lx, ly, tx, ty)
                assert t.size > 0
                d = np.abs(t.astype(np.int64) - whole[oy:oy + t.shape[0], ox:ox + t.shape[1]].astype(np.int64))
                worst, n_diff, n_px = max(worst, int(d.max())), n_diff + int((d > 0).sum()), n_px + d.size
                n_gutter += d.size - min(512, whole.shape[0] - 512 * ty) * min(512, whole.shape[1] - 512 * tx)
    assert n_gutter > 0  # the comparison did include gutter pixels
    assert worst <= 1 and n_diff <= 1e-3 * n_px, (worst, n_diff, n_px)


def test_neighbouring_tiles_agree_on_their_shared_pixels(images):
    """The gutter of tile (tx, 0) holds the same LOD pixels as the core of tile (tx +- 1, 0)."""
    img = images["spectrogram"]
    for lx, ly in [(1, 0), (1, 1)]:
        a, (oxa, _) = orc.spectrogram_tile_u16(img, lx, ly, 0, 0)
        b, (oxb, _) = orc.spectrogram_tile_u16(img, lx, ly, 1, 0)
        assert (oxa, oxb) == (0, 508) and a.shape[1] == 516
        assert np.abs(a[:, 508:516].astype(int) - b[:, 0:8].astype(int)).max() <= 1


# (u16 worst step, u16 mismatch rate, colour-index mismatch rate) allowed per resampler variant; the colour index never
# moves by more than one.  Observed values (worst level of LEVELS) in the comments.
BOUNDS = {
    # spectrogram-like and noise images: nothing sits on a rounding tie
    "spectrogram": {orc.RESIZE_F64_PRENORM: (1, 1e-4, 1e-5),   # 0, 0, 0
                    orc.RESIZE_FIXED_MAX: (1, 2e-4, 1e-5),     # 1, 5.7e-5, 0
                    orc.RESIZE_FIXED_16: (10, 1.0, 1e-2)},     # 8, 0.52, 3.4e-3
    "noise": {orc.RESIZE_F64_PRENORM: (1, 1e-4, 1e-5),         # 0, 0, 0
              orc.RESIZE_FIXED_MAX: (1, 2e-4, 1e-5),           # 1, 3.4e-5, 0
              orc.RESIZE_FIXED_16: (10, 1.0, 1e-2)},           # 6, 0.80, 4.8e-3
    # a two-valued image (0 / 65535) with box-aligned edges: symmetric tap sets put exactly half of the weight on 65535, the
    # exact result is 32767.5 — a rounding tie that is ALSO the boundary between colour indices 128 and 129 — and the sign of
    # the last-bit error decides.  The adversarial end of the scale: any two correct implementations disagree here.
    "steps": {orc.RESIZE_F64_PRENORM: (1, 6e-2, 5e-2),         # 1, 4.1e-2, 2.7e-2
              orc.RESIZE_FIXED_MAX: (1, 6e-2, 5e-2),           # 1, 3.9e-2, 2.7e-2
              orc.RESIZE_FIXED_16: (10, 1.0, 5e-2)},           # 6, 0.50, 2.7e-2
}


@pytest.mark.parametrize("name", ["spectrogram", "noise", "steps"])
def test_fixed_point_lanczos_sensitivity_bound(images, name):
    """How far a fixed-point (or differently rounded f64) Lanczos3 of fast_image_resize's structure can sit from the
    oracle's f64 filter, on the u16 plane and on the colour-index plane — BOUNDS above.  A sensitivity bound, not parity."""
    img = images[name]
    for lx, ly in LEVELS:
        ref = orc.resize_whole_image(img, lx, ly, orc.RESIZE_F64)
        for mode, (u16_worst, u16_rate, ci_rate) in BOUNDS[name].items():
            got = orc.resize_whole_image(img, lx, ly, mode)
            d = np.abs(got.astype(np.int64) - ref.astype(np.int64))
            dc = np.abs(_ci(got) - _ci(ref))
            assert d.max() <= u16_worst and (d > 0).mean() <= u16_rate, (name, lx, ly, mode, d.max(), (d > 0).mean())
            assert dc.max() <= 1 and (dc > 0).mean() <= ci_rate, (name, lx, ly, mode, dc.max(), (dc > 0).mean())


def test_fixed_point_tiles_equal_fixed_point_whole_image(images):
    """Position purity holds for the fixed-point filter too, except that its precision is chosen from the largest
    coefficient of the crop's own table — so tiles may differ from the whole-image resize by one u16 step."""
    img = images["spectrogram"]
    for lx, ly in [(1, 0), (2, 1)]:
        whole = orc.resize_whole_image(img, lx, ly, orc.RESIZE_FIXED_MAX)
        for tx in range(-(-whole.shape[1] // 512)):
            t, (ox, oy) = orc.spectrogram_tile_u16(img, lx, ly, tx, 0, orc.RESIZE_FIXED_MAX)
            d = np.abs(t.astype(np.int64) - whole[oy:oy + t.shape[0], ox:ox + t.shape[1]].astype(np.int64))
            assert d.max() <= 1 and (d > 0).mean() <= 1e-3
