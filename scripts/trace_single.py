#!/usr/bin/env python3
"""Development tool: run the single-track (BASELINE config 2) step a few times; meant to be run under
`rocprofv3 --kernel-trace` to see the kernel timeline of one step (scripts/trace_single.sh)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402
import thesia_amd as ta  # noqa: E402

dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
cmap = open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
sr = 48000
w1 = bench.Workload(torch, ta, ctx, dev, [0], sr, 60 * sr, 2048, 512, 2048, int(os.environ.get("KERNEL", "0")), cmap)
for _ in range(int(os.environ.get("STEPS", "6"))):
    w1.step(None)
torch.cuda.synchronize(dev)
