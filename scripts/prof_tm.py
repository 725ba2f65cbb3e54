#!/usr/bin/env python3
"""Development tool: the TrackManager's re-quantise path (set_dB_range) for 32 tracks, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta
from tests.synth import synth_track
sr, n, n_tracks = 48000, 48000 * 30, 32
cmap = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
host = np.stack([synth_track(i, sr, n) for i in range(4)])
with ta.Context(0) as ctx:
    tm = ta.TrackManager(ctx)
    tm.set_setting(2048 / 48, 4, 1, ta.LINEAR)
    tm.set_colormap(cmap)
    tm.add_tracks([(i, sr, host[i % 4][None]) for i in range(n_tracks)])
    tm.apply_track_list_changes()
    for r in (80.0, 100.0, 90.0):
        t0 = time.perf_counter()
        tm.set_dB_range(r)
        print(f"set_dB_range: {(time.perf_counter() - t0) * 1e3:.2f} ms")
    tm.close()
