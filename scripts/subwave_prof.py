#!/usr/bin/env python3
"""Development tool: per-phase shader-clock totals of stft_subwave_kernel (a -DTH_SUBW_PROF=k build writes phase k's ticks per chunk
into the chunk's min slot and the loop total into its max slot; wave_post folds them per channel: min of the phase, max of the total).
usage: THESIA_AMD_LIB=scripts/variants/libthesia_amd_subw_prof<k>.so python scripts/subwave_prof.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import thesia_amd as ta
from bench import synth_on_gpu
dev = torch.device("cuda", 0)
ctx = ta.Context(0)
sr, n_fft, hop, tracks = 48000, 32768, 8192, 128
n = 30 * sr
wav = synth_on_gpu(torch, dev, list(range(tracks)), sr, n)
plan = ta.Plan(ctx, sr, n_fft, hop, n_fft, ta.LINEAR)
T, H = plan.n_frames(n), plan.height
sp = ta.pitch_f32(H)
spec = torch.empty((tracks, T, sp), dtype=torch.float32, device=dev)
mm = torch.empty((tracks, 2), dtype=torch.float32, device=dev)
chan = (ta.ChanDesc * tracks)(*[ta.ChanDesc(wav[i].data_ptr(), spec[i].data_ptr(), n, T, sp) for i in range(tracks)])
for _ in range(3):
    plan.calc_spec_batch_dev(chan, mm.data_ptr())
torch.cuda.synchronize()
m = mm.cpu().numpy()
print(f"ticks per chunk (max over a channel's chunks; median over channels) {np.median(m[:, 1]):.0f}")
