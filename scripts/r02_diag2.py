#!/usr/bin/env python3
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta
from oracle import oracle as orc
from tests.synth import synth_track
ctx = ta.Context(0)
win, hop, n_fft = 2048, 512, 2048
n = 40000 + win
x = synth_track(n_fft + hop, 48000, n)
want, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
plan = ta.Plan(ctx, 48000, win, hop, n_fft, ta.LINEAR)
plan.set_kernel(2)
spec, mn, mx = plan.calc_spec(x)
ga = np.power(10.0, spec.astype(np.float64) / 20.0)
fm = amp.max(axis=1, keepdims=True)
rel = np.abs(ga - amp) / fm
for f in range(amp.shape[0]):
    b = np.nonzero(rel[f] > 2e-6)[0]
    if len(b):
        print(f"frame {f}: {len(b)} bad bins; mod16 set {sorted(set((b % 16).tolist()))}; mod64 set {sorted(set((b % 64).tolist()))[:10]}; first {b[:6].tolist()} got {ga[f, b[:3]]} want {amp[f, b[:3]]}")
