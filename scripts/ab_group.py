#!/usr/bin/env python3
"""Development tool (round 3): A/B of the wave kernel's chunk schedules on one box — the default (independent waves, long
chunks from the device-wide queue) against the workgroup-level group schedule at several chunk lengths; checks that the two
produce identical spectrograms, then alternates timed launches.  usage: python scripts/ab_group.py [--reps 30] [--gap-ms 1]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import thesia_amd as ta  # noqa: E402
from bench import synth_on_gpu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--gap-ms", type=float, default=1.0)
ap.add_argument("--tracks", type=int, default=128)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--variants", type=str, default="0:0,4:1,6:1,8:1,12:1,16:1,8:0")
ap.add_argument("--waves", type=int, default=0)
a = ap.parse_args()
sr, win, hop, n_fft = 48000, 2048, 512, 2048
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
n = int(a.seconds * sr)
wav = synth_on_gpu(torch, dev, list(range(a.tracks)), sr, n)
torch.cuda.synchronize()
variants = [tuple(int(v) for v in s.split(":")) for s in a.variants.split(",")]
plans, specs = [], []
T = H = sp = None
for chunk, group in variants:
    plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.LINEAR)
    plan.set_kernel(2 | (a.waves << 8) | (chunk << 16) | (group << 24))
    T, H = plan.n_frames(n), plan.height
    sp = ta.pitch_f32(H)
    plans.append(plan)
spec = torch.empty((a.tracks, T, sp), dtype=torch.float32, device=dev)
ref = None
mm = torch.empty((a.tracks, 2), dtype=torch.float32, device=dev)
chan = (ta.ChanDesc * a.tracks)(*[ta.ChanDesc(wav[i].data_ptr(), spec[i].data_ptr(), n, T, sp) for i in range(a.tracks)])
for (chunk, group), plan in zip(variants, plans):
    spec.fill_(float("nan"))
    plan.calc_spec_batch_dev(chan, mm.data_ptr())
    torch.cuda.synchronize()
    if ref is None:
        ref, ref_mm = spec.clone(), mm.clone()
    else:
        same = torch.equal(torch.nan_to_num(spec, nan=-1.0), torch.nan_to_num(ref, nan=-1.0)) and torch.equal(mm, ref_mm)
        print(f"chunk {chunk} group {group}: identical to the default schedule: {same}", flush=True)
times = [[] for _ in variants]
for r in range(a.reps + 3):
    for k, plan in enumerate(plans):
        if a.gap_ms > 0:
            time.sleep(a.gap_ms * 1e-3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.calc_spec_batch_dev(chan, mm.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        if r >= 3:
            times[k].append(e0.elapsed_time(e1))
frames = a.tracks * T
for (chunk, group), ts in zip(variants, times):
    ms = float(np.median(ts))
    print(f"chunk {chunk:2d} group {group}: median {ms:.3f} ms  min {min(ts):.3f}  p90 {np.percentile(ts, 90):.3f}   "
          f"{frames * 6148 / ms / 1e6 / 80:.1f} % of 8 TB/s", flush=True)
