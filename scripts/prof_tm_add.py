#!/usr/bin/env python3
"""Development tool: phase timing of th_tm_add_tracks / apply_track_list_changes for 32 tracks x 30 s from pageable host memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta
from tests.synth import synth_track
sr, n, n_tracks = 48000, 48000 * 30, 32
cmap = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
base = np.stack([synth_track(i, sr, n) for i in range(4)])
host = np.stack([base[i % 4] for i in range(n_tracks)]).copy()
with ta.Context(0) as ctx:
    for rep in range(3):
        tm = ta.TrackManager(ctx)
        tm.set_setting(2048 / 48, 4, 1, ta.LINEAR)
        tm.set_colormap(cmap)
        t0 = time.perf_counter()
        tm.add_tracks([(i, sr, host[i][None]) for i in range(n_tracks)])
        t1 = time.perf_counter()
        tm.apply_track_list_changes()
        t2 = time.perf_counter()
        frames = n_tracks * ta.stft_n_frames(n, 2048, 512)
        print(f"rep {rep}: add_tracks {(t1 - t0) * 1e3:.2f} ms, apply_track_list_changes {(t2 - t1) * 1e3:.2f} ms -> {frames / (t2 - t0) / 1e6:.2f} M frames/s compute-only", flush=True)
        tm.close()
