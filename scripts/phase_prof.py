#!/usr/bin/env python3
"""Development tool: per-phase shader-clock breakdown of the wave STFT kernel's frame loop.
Build the instrumented variant first:  patch -p1 < scripts/patches/instrumentation_phase_prof_wave_times.patch; scripts/build_variant.sh prof -DTH_PHASE_PROF; patch -R -p1 < (same)
run:  THESIA_AMD_LIB=scripts/variants/libthesia_amd_prof.so python scripts/phase_prof.py [--nfft 2048]"""
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
ap.add_argument("--nfft", type=int, default=2048)
ap.add_argument("--win", type=int, default=0)
ap.add_argument("--hop", type=int, default=0)
ap.add_argument("--kernel", type=int, default=0)
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
if a.kernel:
    plan.set_kernel(a.kernel)
T, H = plan.n_frames(n), plan.height
sp = ta.pitch_f32(H)
spec = torch.empty((a.tracks, T, sp), dtype=torch.float32, device=dev)
mm = torch.empty((a.tracks, 2), dtype=torch.float32, device=dev)
chan = (ta.ChanDesc * a.tracks)(*[ta.ChanDesc(wav[i].data_ptr(), spec[i].data_ptr(), n, T, sp) for i in range(a.tracks)])
for _ in range(3):
    plan.calc_spec_batch_dev(chan, mm.data_ptr())
torch.cuda.synchronize()
fn = _ffi.lib.th_debug_phase_prof
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
names = ["window+cursor+fetch issue", "pass1 + LDS writes", "read1 (exchange-1 reads)", "pass2 + LDS writes", "read2 (+flush check)",
         "pass3", "split + dB + stores", "(exp) before vmcnt(0) wait", "loop top (cur = nxt)"]
frames = a.tracks * T
tot = sum(buf[i] for i in range(9))
n_waves = 256 * (((a.kernel >> 8) & 0xFF) or 12)
print(f"{ms:.3f} ms for {frames} frames (instrumented); ticks are s_memtime units summed over all waves")
for i, nm in enumerate(names):
    if not buf[i]:
        continue
    print(f"  {nm:28s} {buf[i] / frames:9.1f} ticks/frame  {100.0 * buf[i] / tot:5.1f} %")
if buf[9]:
    print(f"  shader clock during the frame loop: {buf[10] / buf[9] * 100.0:.0f} MHz (s_memtime ticks per 100 MHz s_memrealtime tick); loop wall time per wave {buf[9] / n_waves / 100.0:.1f} us")
print(f"  {'total':28s} {tot / frames:9.1f} ticks/frame;  wave-time {tot / n_waves / 1e6:.3f} Mticks per wave -> {tot / n_waves / (ms * 1e-3) / 1e6:.1f} MHz tick rate if waves were busy the whole launch")
