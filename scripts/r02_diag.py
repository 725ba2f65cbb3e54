#!/usr/bin/env python3
"""Development tool: wave kernel vs generic kernel vs oracle on a few framings; prints where they differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta
from oracle import oracle as orc
from tests.synth import synth_track

ctx = ta.Context(0)
for win, hop, n_fft in [(2048, 512, 2048), (1920, 480, 2048), (1764, 441, 2048), (2048, 1024, 2048), (2048, 256, 2048), (2047, 2047, 2048)]:
    n = 40000 + win
    x = synth_track(n_fft + hop, 48000, n)
    want, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
    out = {}
    for which in (1, 2):
        plan = ta.Plan(ctx, 48000, win, hop, n_fft, ta.LINEAR)
        plan.set_kernel(which)
        spec, mn, mx = plan.calc_spec(x)
        out[which] = spec
        plan.close()
    for which in (1, 2):
        ga = np.power(10.0, out[which].astype(np.float64) / 20.0)
        fm = amp.max(axis=1, keepdims=True)
        rel = np.abs(ga - amp) / fm
        bad = np.argwhere(rel > 2e-6)
        print(f"{win}/{hop}/{n_fft} kernel {which}: max rel {rel.max():.3e}; bad frames {sorted(set(bad[:,0].tolist()))[:20]} "
              f"of {amp.shape[0]}; bad bins (first bad frame) {bad[bad[:,0]==bad[0,0]][:12,1].tolist() if len(bad) else []}", flush=True)
