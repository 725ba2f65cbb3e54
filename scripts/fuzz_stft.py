#!/usr/bin/env python3
"""Development tool: random framings / lengths / batch shapes through the auto-selected kernel (wave kernel with all its
modes: register reuse, phased, dynamic, boundary frames, fused mel) against the generic kernel, which shares none of that
code.  usage: python scripts/fuzz_stft.py [seconds] [seed] [big]     (big: long channels, many chunks per wave)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
ctx = ta.Context(0)
t_end = time.time() + budget
n_cases = 0
worst = 0.0
while time.time() < t_end:
    n_fft = int(rng.choice([1024, 2048, 2048, 4096]))
    win = int(rng.integers(n_fft // 2 + 1, n_fft + 1))
    if rng.random() < 0.5:
        win = n_fft if rng.random() < 0.5 else win // 2 * 2
    hop = int(rng.choice([win // 4, win // 2, win // 8, int(rng.integers(1, win + 1)), 480, 441, 160, 221, 320]))
    hop = max(1, min(hop, win))
    mel = rng.random() < 0.35
    n_mel = int(rng.choice([0, 40, 128, 200])) if mel else 0
    sr = int(rng.choice([16000, 22050, 44100, 48000]))
    try:
        plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.MEL if mel else ta.LINEAR, n_mel)
        ref = ta.Plan(ctx, sr, win, hop, n_fft, ta.MEL if mel else ta.LINEAR, n_mel)
    except ta.ThError:
        continue
    ref.set_kernel(1)
    big = len(sys.argv) > 3
    lens = [int(rng.integers(1, 6 * n_fft)) for _ in range(int(rng.integers(1, 5)))] + [int(rng.integers(n_fft, 40 * n_fft))]
    if big:
        lens += [int(rng.integers(200 * n_fft, 1500 * n_fft)) for _ in range(int(rng.integers(1, 4)))]
        hop = max(hop, 64)
    if hop < 8:
        lens = [min(v, 3 * n_fft) for v in lens]
    wavs = [(rng.standard_normal(v) * 0.1 + 0.3 * np.sin(np.arange(v) * rng.uniform(0.001, 1.0))).astype(np.float32) for v in lens]
    a, mma = plan.calc_spec_batch(wavs)
    b, mmb = ref.calc_spec_batch(wavs)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape, (win, hop, n_fft, lens[i])
        ax, ay = np.power(10.0, x.astype(np.float64) / 20), np.power(10.0, y.astype(np.float64) / 20)
        scale = np.maximum(ay.max(axis=1, keepdims=True), 1e-30)
        err = float((np.abs(ax - ay) / scale).max()) if x.size else 0.0
        worst = max(worst, err)
        assert err <= 5e-6, (plan.kernel_name, win, hop, n_fft, n_mel, lens[i], err)
        assert mma[i, 0] == x.min() and mma[i, 1] == x.max(), (plan.kernel_name, win, hop, n_fft, lens[i])
    plan.close()
    ref.close()
    n_cases += 1
print(f"{n_cases} random cases, worst |X| difference between the two kernels {worst:.2e} of the frame maximum")
