#!/usr/bin/env python3
"""Development tool: per-phase shader-clock breakdown of the workgroup-per-frame STFT kernel (n_fft 8192 ... 32768), as seen by
thread 0 of every workgroup.  Build the instrumented variant first:
  patch -p1 < scripts/patches/instrumentation_block_prof.patch; scripts/build_variant.sh bprof -DTH_BLOCK_PROF; patch -R -p1 < (same)
run:  THESIA_AMD_LIB=scripts/variants/libthesia_amd_bprof.so python scripts/block_prof.py --nfft 32768 [--win W --hop H]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import thesia_amd as ta  # noqa: E402
from thesia_amd import _ffi  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=128)
ap.add_argument("--seconds", type=float, default=30.0)
ap.add_argument("--nfft", type=int, default=32768)
ap.add_argument("--win", type=int, default=0)
ap.add_argument("--hop", type=int, default=0)
a = ap.parse_args()
sr, n_fft = 48000, a.nfft
win = a.win or n_fft
hop = a.hop or win // 4
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
n = int(a.seconds * sr)
wav = (torch.rand((a.tracks, n), device=dev) * 2 - 1) * 0.3
plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.LINEAR)
T, H = plan.n_frames(n), plan.height
sp = ta.pitch_f32(H)
spec = torch.empty((a.tracks, T, sp), dtype=torch.float32, device=dev)
mm = torch.empty((a.tracks, 2), dtype=torch.float32, device=dev)
chan = (ta.ChanDesc * a.tracks)(*[ta.ChanDesc(wav[i].data_ptr(), spec[i].data_ptr(), n, T, sp) for i in range(a.tracks)])
for _ in range(3):
    plan.calc_spec_batch_dev(chan, mm.data_ptr())
torch.cuda.synchronize()
fn = _ffi.lib.th_debug_block_prof
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
buf = (ctypes.c_ulonglong * 16)()
assert fn(buf, 1) == 0
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
plan.calc_spec_batch_dev(chan, mm.data_ptr())
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
assert fn(buf, 0) == 0
names = ["fetch + window multiply (incl. waiting for the samples)", "first pass + LDS stores", "barriers (all of the frame's)", "exchange reads (all)",
         "pass A + stores", "pass B + stores", "last pass + Z stores", "split pass + dB + row stores issued"]
frames = buf[8]
tot = sum(buf[i] for i in range(8))
print(f"{plan.kernel_name} n_fft {n_fft} win {win} hop {hop}: {ms:.3f} ms (instrumented), {frames} interior frames in the block kernel; "
      f"shader-clock ticks per frame as thread 0 of the workgroup sees them")
for i, nm in enumerate(names):
    print(f"  {nm:58s} {buf[i] / max(frames, 1):9.0f}   {100.0 * buf[i] / max(tot, 1):5.1f} %")
print(f"  {'total':58s} {tot / max(frames, 1):9.0f}   = {tot / max(frames, 1) / 2.4e3:.2f} us per frame at 2.4 GHz; "
      f"launch: {ms * 1e3 * 256 / max(frames, 1):.2f} us per frame and CU")
