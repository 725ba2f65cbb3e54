#!/usr/bin/env python3
"""Test tool (under tests/ because the oracle is the checker; not collected by pytest): random channel lengths and level
counts through th_waveform_pyramid_dev and the per-tile encoder, against the oracle's encode_waveform_tile
(render_tiles.rs:232-279): min / max bit-exact, mean bit-exact for bins <= 16 samples, 1e-6 of the peak above.
usage: python tests/fuzz_waveform.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import thesia_amd as ta  # noqa: E402
from oracle import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ta.Context(0)
t_end = time.time() + budget
n_cases = 0
while time.time() < t_end:
    n = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(1, 9000)), int(rng.integers(4000, 300000)),
                        4096 * int(rng.integers(1, 40)) + int(rng.integers(-2, 3))]))
    n = max(1, n)
    x = (rng.standard_normal(n) * rng.uniform(0.01, 0.5)).astype(np.float32)
    n_levels = int(rng.integers(1, 19))
    lv = ctx.waveform_pyramid(x, n_levels)
    peak = float(np.abs(x).max())
    for level in range(n_levels):
        bins = -(-n // (1 << level))
        assert lv[level].shape == (bins, 3), (n, level)
        n_tiles = -(-bins // 1024)
        for t in sorted({0, n_tiles - 1, int(rng.integers(0, n_tiles))}):
            want = np.frombuffer(orc.encode_waveform_tile(x, 1, level, t)[24:], np.float32).reshape(-1, 3)
            got = lv[level][1024 * t: 1024 * (t + 1)]
            assert got.shape == want.shape and np.array_equal(got[:, :2], want[:, :2]), (n, level, t)
            if level <= 4:
                assert np.array_equal(got[:, 2], want[:, 2]), (n, level, t)
            else:
                assert np.abs(got[:, 2] - want[:, 2]).max() <= 1e-6 * max(peak, 1e-30), (n, level, t)
            if n_cases % 8 == 0:  # the per-tile kernel as well (headers included)
                assert ctx.encode_waveform_tile(x, 1, level, t)[:24] == orc.encode_waveform_tile(x, 1, level, t)[:24]
    n_cases += 1
print(f"{n_cases} random cases: every pyramid level equals the oracle's tile bins")
