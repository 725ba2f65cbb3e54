#!/usr/bin/env python3
"""Test tool (lives under tests/ because it uses the oracle as the checker; not collected by pytest): random shapes / row ranges / dB ranges / colormap lengths through the quantiser and the level-0 tile
encoder, bit for bit against the oracle (drawing.rs:4-33, render_tiles.rs:281-352).
usage: python tests/fuzz_img.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import thesia_amd as ta  # noqa: E402
from oracle import oracle as orc  # noqa: E402  (development tool: the checker, as in tests/)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ta.Context(0)
cmap = open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
t_end = time.time() + budget
n_cases = n_lod = 0
while time.time() < t_end:
    T, H = int(rng.integers(1, 1400)), int(rng.choice([1, 2, 40, 127, 128, 129, 130, 257, 513, 1025, int(rng.integers(1, 1200))]))
    spec = rng.uniform(-150, 20, (T, H)).astype(np.float32)
    k = max(1, spec.size // 500)
    spec.ravel()[rng.integers(0, spec.size, k)] = -np.inf
    spec.ravel()[rng.integers(0, spec.size, k)] = np.nan
    spec.ravel()[rng.integers(0, spec.size, k)] = np.inf
    lo = float(np.float32(rng.uniform(-140, -20)))
    hi = float(np.float32(lo + rng.choice([0.0, 1e-3, 1.0, 37.5, 60.0, 100.0, 159.99])))
    # quantisation boundaries: values whose scaled position is exactly k + 0.5
    cm = int(rng.choice([2, 3, 4, 16, 256, 258, 1024, 65536]))
    i0 = int(rng.integers(0, H))
    i1 = int(rng.integers(i0 + 1, H + int(rng.integers(1, 40))))  # rows >= H are zero-filled (mixed sample rates)
    got = ctx.spec_to_img(spec, (i0, i1), (lo, hi), cm)
    want = orc.convert_spectrogram_to_img(spec, (i0, i1), (lo, hi), cm)
    assert np.array_equal(got, want), (T, H, i0, i1, lo, hi, cm, int((got != want).sum()))
    if n_cases % 4 == 0 and got.shape[0] >= 1:
        img = got
        W_, H_ = img.shape[1], img.shape[0]
        tx, ty = int(rng.integers(0, -(-W_ // 512))), int(rng.integers(0, -(-H_ // 512)))
        a = ctx.encode_spectrogram_tile(img, cmap, 5, 0, 0, tx, ty)
        b = orc.encode_spectrogram_tile(img, cmap, 5, 0, 0, tx, ty)
        assert a == b, (W_, H_, tx, ty)
    if n_cases % 16 == 0 and got.shape[0] >= 2 and got.shape[1] >= 2:  # LOD > 0: the restated separable Lanczos3
        lx, ly = int(rng.integers(0, 4)), int(rng.integers(0, 4))
        if lx or ly:
            wl, hl = -(-got.shape[1] >> lx) if lx else got.shape[1], -(-got.shape[0] >> ly) if ly else got.shape[0]
            wl, hl = -(-got.shape[1] // (1 << lx)), -(-got.shape[0] // (1 << ly))
            tx, ty = int(rng.integers(0, -(-wl // 512))), int(rng.integers(0, -(-hl // 512)))
            a = ctx.encode_spectrogram_tile(got, cmap, 9, lx, ly, tx, ty)
            b = orc.encode_spectrogram_tile(got, cmap, 9, lx, ly, tx, ty)
            assert a == b, ("lod", got.shape, lx, ly, tx, ty)
            n_lod += 1
    n_cases += 1
print(f"{n_cases} random cases ({n_lod} LOD tiles): u16 images and tiles bit-identical to the oracle")
