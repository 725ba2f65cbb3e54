#!/usr/bin/env python3
"""Development tool: random mel banks at n_fft 512 ... 16384 — the moment-form epilogue (default) against the two-kernel route
(selector 12: the reference's f32 table) on the same noise + lone lines; prints the worst difference relative to the frame maximum."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta  # noqa: E402

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = np.random.default_rng(seed)
ctx = ta.Context(0)
t_end = time.time() + budget
worst, n_cases, n_fused = 0.0, 0, 0
while time.time() < t_end:
    n_fft = int(rng.choice([512, 1024, 2048, 4096, 4096, 8192, 16384]))
    sr = int(rng.choice([8000, 16000, 22050, 32000, 44100, 48000, 88200, 96000, 192000]))
    n_mel = int(rng.choice([0, int(rng.integers(1, 64)), int(rng.integers(64, 1200)), int(rng.integers(min(1200, n_fft // 4), max(1201, n_fft // 2 + 300))), int(rng.integers(n_fft // 2, 3 * n_fft))]))
    hop = n_fft // 4 if rng.random() < 0.5 else int(rng.integers(max(1, n_fft // 16), n_fft))
    win = n_fft if rng.random() < 0.6 else int(rng.integers(n_fft // 2 + 1, n_fft + 1))
    try:
        plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.MEL, n_mel)
    except ta.ThError as e:
        print(f"refused sr {sr} n_fft {n_fft} n_mel {n_mel}: {e}")
        continue
    info = plan.mel_moments_info()
    x = (rng.standard_normal(5 * n_fft + int(rng.integers(0, 999))) * 0.1).astype(np.float32)
    k = rng.integers(0, len(x), 8)
    x[k] += 0.5
    a, _, _ = plan.calc_spec(x)
    name = plan.kernel_name
    plan.set_kernel(12)
    b, _, _ = plan.calc_spec(x)
    plan.close()
    n_cases += 1
    fused = info["groups"] != 0 and "fused" in name
    n_fused += bool(fused)
    la, lb = 10.0 ** (a.astype(np.float64) / 20), 10.0 ** (b.astype(np.float64) / 20)
    fin = np.isfinite(a) == np.isfinite(b)
    assert fin.all(), (sr, n_fft, win, hop, n_mel)
    d = (np.abs(la - lb).max(axis=1) / np.maximum(lb.max(axis=1), 1e-300)).max()
    worst = max(worst, d)
    if d > 2e-5:
        print(f"LARGE sr {sr} n_fft {n_fft} win {win} hop {hop} n_mel {n_mel} ({name}; {info}): {d:.3e}", flush=True)
print(f"seed {seed}: {n_cases} banks ({n_fused} with a moment table), worst difference {worst:.3e} of the frame maximum")
