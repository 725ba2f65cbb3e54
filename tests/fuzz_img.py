#!/usr/bin/env python3
"""Command-line form of tests/fuzzers.py::fuzz_img (test tool: the oracle is the checker).
usage: python tests/fuzz_img.py [seconds] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import thesia_amd as ta  # noqa: E402
from tests import fuzzers  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print(fuzzers.fuzz_img(ta.Context(0), seed=seed, max_seconds=budget))
