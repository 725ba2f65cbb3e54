#!/usr/bin/env python3
"""STFT-kernel microbenchmark (development tool): the bench.py workload, STFT->dB launch only.
usage: python scripts/bench_stft.py [--kernel K] [--tracks N] [--seconds S] [--reps R] [--nfft 2048]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import thesia_amd as ta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--kernel", type=int, nargs="*", default=[0])
ap.add_argument("--tracks", type=int, default=128)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--nfft", type=int, default=2048)
ap.add_argument("--win", type=int, default=0)
ap.add_argument("--hop", type=int, default=0)
ap.add_argument("--gap-ms", type=float, default=0.0, help="idle time between launches (power-limited kernel: "
                "back-to-back launches run slower than launches with pauses or lighter kernels in between)")
ap.add_argument("--mel", type=int, default=None, help="mel plan with this many bins (0 = the reference's default count)")
ap.add_argument("--sr", type=int, default=48000)
ap.add_argument("--noise", action="store_true", help="full-scale white noise instead of the bench's SURVEY 8(d) tracks "
                "(the kernel is power-limited: noise costs ~30 %% more time for the same work)")
a = ap.parse_args()
sr = a.sr
n_fft = a.nfft
win = a.win or n_fft
hop = a.hop or win // 4
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
n = int(a.seconds * sr)
g = torch.Generator(device=dev)
g.manual_seed(1)
if a.noise:
    wav = (torch.rand((a.tracks, n), device=dev, generator=g) * 2 - 1) * 0.3
else:
    from bench import synth_on_gpu
    wav = synth_on_gpu(torch, dev, list(range(a.tracks)), sr, n)
    torch.cuda.synchronize()
for K in a.kernel:
    plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.LINEAR) if a.mel is None else ta.Plan(ctx, sr, win, hop, n_fft, ta.MEL, a.mel)
    if K:
        plan.set_kernel(K)
    T, H = plan.n_frames(n), plan.height
    sp = H if os.environ.get('TH_DENSE') == '1' else ta.pitch_f32(H)
    spec = torch.empty((a.tracks, T, sp), dtype=torch.float32, device=dev)
    mm = torch.empty((a.tracks, 2), dtype=torch.float32, device=dev)
    chan = (ta.ChanDesc * a.tracks)(*[ta.ChanDesc(wav[i].data_ptr(), spec[i].data_ptr(), n, T, sp) for i in range(a.tracks)])
    for _ in range(3):
        plan.calc_spec_batch_dev(chan, mm.data_ptr())
    torch.cuda.synchronize()
    ts = []
    import time
    for _ in range(a.reps):
        if a.gap_ms > 0:
            time.sleep(a.gap_ms * 1e-3)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.calc_spec_batch_dev(chan, mm.data_ptr())
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ms = float(np.median(ts))
    frames = a.tracks * T
    bpf = 4 * hop + 4 * H
    print(f"kernel={K} ({plan.kernel_name}) n_fft={n_fft} win={win} hop={hop}: median {ms:.3f} ms  min {min(ts):.3f} ms  "
          f"{frames / ms / 1e3:.1f} Mframes/s  {frames * bpf / ms / 1e6:.0f} GB/s algorithmic "
          f"({frames * bpf / ms / 1e6 / 80:.1f}% of 8 TB/s)", flush=True)
    del spec
