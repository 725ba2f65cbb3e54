#!/usr/bin/env python3
"""Test tool: a longer soak of tests/fuzzers.py with seeds other than the ones pytest uses (the oracle is the checker).
usage: python scripts/fuzz_soak.py [seconds per fuzzer and seed] [first seed] [number of seeds]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import thesia_amd as ta  # noqa: E402
from tests import fuzzers  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 100
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
with ta.Context(0) as ctx:
    for seed in range(seed0, seed0 + n_seeds):
        print(seed, "stft", fuzzers.fuzz_stft(ctx, seed=seed, max_seconds=budget, cap_bytes=1 << 30), flush=True)
        print(seed, "stft big", fuzzers.fuzz_stft(ctx, seed=seed + 1000, max_seconds=budget, big=True, cap_bytes=1 << 30), flush=True)
        print(seed, "track manager", fuzzers.fuzz_track_manager(ctx, seed=seed, max_seconds=budget), flush=True)
        print(seed, "img", fuzzers.fuzz_img(ctx, seed=seed, max_seconds=budget / 2), flush=True)
        print(seed, "waveform", fuzzers.fuzz_waveform(ctx, seed=seed, max_seconds=budget / 2), flush=True)
print("soak ok")
