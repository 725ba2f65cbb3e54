#!/usr/bin/env python3
"""Development tool: error against the oracle's f64 DFT and launch time of the chirp-z plans (odd factor of n_fft above 63)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import thesia_amd as ta  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from synth import synth_track  # noqa: E402

ctx = ta.Context(0)
for sr, win_ms, t_ov, f_ov in [(8000, 2.0, 4, 65), (16000, 8.0, 2, 71), (48000, 40.0, 4, 67), (48000, 40.0, 4, 3), (48000, 40.0, 4, 63)]:
    hop, win, n_fft = ta.calc_framing_params(win_ms, t_ov, f_ov, sr)
    plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.LINEAR)
    x = synth_track(7, sr, 3 * n_fft + 17)
    spec, _, _ = plan.calc_spec(x)
    want, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
    got_amp = 10.0 ** (spec.astype(np.float64) / 20)
    rel = (np.abs(got_amp - amp).max(axis=1) / amp.max(axis=1)).max()
    big = amp >= 1e-2 * amp.max(axis=1, keepdims=True)
    ddb = np.abs(spec.astype(np.float64) - want)[big].max()
    t0 = time.time()
    for _ in range(3):
        plan.calc_spec(x)
    dt = (time.time() - t0) / 3
    print(f"sr {sr} f_overlap {f_ov}: n_fft {n_fft} ({plan.kernel_name}), {spec.shape[0]} frames: max error {rel:.2e} of the frame maximum, "
          f"{ddb:.2e} dB where >= 1 % of it; {dt * 1e3:.1f} ms per call incl. transfers", flush=True)
    plan.close()
