#!/usr/bin/env python3
"""Development tool: keep one kind of work running for some seconds (to read the package power next to it).
usage: power_loops.py copy|store|read|stft [seconds]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
kind, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 12.0
dev = torch.device("cuda", 0)
n = 1 << 28
a = torch.empty(n, dtype=torch.float32, device=dev).normal_()
b = torch.empty_like(a)
t0 = time.time(); it = 0
torch.cuda.synchronize()
while time.time() - t0 < secs:
    for _ in range(20):
        if kind == "copy": b.copy_(a)
        elif kind == "store": b.fill_(1.5)
        elif kind == "read": s = a.sum()
    torch.cuda.synchronize(); it += 20
dt = time.time() - t0
byt = {"copy": 2, "store": 1, "read": 1}[kind] * n * 4
print(f"{kind}: {it} iterations, {byt * it / dt / 1e12:.2f} TB/s")
