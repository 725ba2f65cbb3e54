#!/usr/bin/env python3
"""Development tool: when does every wave of the wave STFT kernel start its frame loop and when does it finish?
Build the instrumented variant first:  patch -p1 < scripts/patches/instrumentation_phase_prof_wave_times.patch; scripts/build_variant.sh wt -DTH_WAVE_TIMES; patch -R -p1 < (same)
run:  THESIA_AMD_LIB=scripts/variants/libthesia_amd_wt.so python scripts/wave_times.py [--nfft 2048]"""
import argparse
import ctypes
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import thesia_amd as ta  # noqa: E402
from thesia_amd import _ffi  # noqa: E402
from bench import synth_on_gpu  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=128)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--nfft", type=int, default=2048)
ap.add_argument("--win", type=int, default=0)
ap.add_argument("--hop", type=int, default=0)
ap.add_argument("--kernel", type=int, default=0)
ap.add_argument("--waves", type=int, default=12)
a = ap.parse_args()
sr, n_fft = 48000, a.nfft
win = a.win or n_fft
hop = a.hop or win // 4
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
n = int(a.seconds * sr)
wav = synth_on_gpu(torch, dev, list(range(a.tracks)), sr, n)
plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.LINEAR)
if a.kernel:
    plan.set_kernel(a.kernel)
T, H = plan.n_frames(n), plan.height
sp = ta.pitch_f32(H)
spec = torch.empty((a.tracks, T, sp), dtype=torch.float32, device=dev)
mm = torch.empty((a.tracks, 2), dtype=torch.float32, device=dev)
chan = (ta.ChanDesc * a.tracks)(*[ta.ChanDesc(wav[i].data_ptr(), spec[i].data_ptr(), n, T, sp) for i in range(a.tracks)])
for _ in range(5):
    plan.calc_spec_batch_dev(chan, mm.data_ptr())
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.calc_spec_batch_dev(chan, mm.data_ptr())
e1.record()
torch.cuda.synchronize()
nw = 256 * a.waves
fn = _ffi.lib.th_debug_wave_times
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * (6 * nw))()
assert fn(buf, 6 * nw) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nw, 6).astype(np.int64)
t0 = t[:, 0].min()
entry, loop, end = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
frames, chunks, start_us = t[:, 3] & 0xFFFF, (t[:, 3] >> 16) & 0xFFFF, (t[:, 3] >> 32) / 100.0
print(f"launch {e0.elapsed_time(e1) * 1000:.1f} us by events; {nw} waves, {frames.sum()} frames; all times in us after the first wave's entry")
q = [0, 1, 10, 50, 90, 99, 100]
for name, v in (("kernel entry", entry), ("frame loop start", loop), ("exit", end), ("loop duration", end - loop),
                ("frames per wave", frames.astype(float)), ("chunks per wave", chunks.astype(float)),
                ("chunk starts, us", start_us), ("us per chunk start", start_us / np.maximum(chunks, 1)), ("us per frame", (end - loop) / np.maximum(frames, 1))):
    print(f"  {name:18s} " + "  ".join(f"p{p}={np.percentile(v, p):8.2f}" for p in q))
busy = (end - loop).sum()
span = end.max()
print(f"  wave-time inside the loops {busy / (nw * span) * 100:.1f} % of waves x kernel span ({span:.1f} us); "
      f"mean idle at the end {np.mean(span - end):.1f} us, before the loop {np.mean(loop):.1f} us")
for x in range(8):
    sel = (np.arange(nw) // a.waves) % 8 == x
    print(f"  XCD {x}: exit p50 {np.percentile(end[sel], 50):7.1f}  max {end[sel].max():7.1f}  frames {frames[sel].sum()}")

# hardware placement: HW_ID bits [3:0] wave slot, [5:4] SIMD, [11:8] CU, [12] SH, [15:13] SE (gfx9 layout); XCC_ID low bits = XCD
hw, xcc = t[:, 4] & 0xFFFFFFFF, (t[:, 4] >> 32) & 0xF
slot, simd, cu, se = hw & 0xF, (hw >> 4) & 3, (hw >> 8) & 0xF, (hw >> 13) & 7
upf = (end - loop) / np.maximum(frames, 1)
print("  us per frame by wave slot on the SIMD:")
for v in sorted(set(slot.tolist())):
    sel = slot == v
    print(f"    slot {v}: n={sel.sum():5d}  us/frame p50 {np.percentile(upf[sel], 50):6.2f}  mean frames {frames[sel].mean():6.1f}")
print("  us per frame by SIMD:", "  ".join(f"{v}: {np.median(upf[simd == v]):.2f}" for v in range(4)))
print("  us per frame by XCC:", "  ".join(f"{v}: {np.median(upf[xcc == v]):.2f}" for v in sorted(set(xcc.tolist()))))
key = (xcc * 8 + se) * 64 + cu * 4 + simd
order = np.argsort(key * 16 + slot)
print("  first SIMDs (xcc se cu simd: slot/us-per-frame ...):")
last, line, shown = None, "", 0
for i in order:
    if key[i] != last:
        if line and shown < 12:
            print("   ", line)
            shown += 1
        last, line = key[i], f"{xcc[i]} {se[i]} {cu[i]:2d} {simd[i]}:"
    line += f"  {slot[i]}/{upf[i]:.2f}/{frames[i]}"
