#!/usr/bin/env python3
"""Test tool (under tests/ because the oracle is the checker; not collected by pytest): random sequences of TrackManager
operations — add_tracks / remove_track / apply_track_list_changes / set_setting / set_dB_range, mixed sample rates and
channel counts — with the state checked against the oracle pipeline after every step (core/mod.rs:62-230): spec
shapes and values, global dB range and max sample rate, every u16 image bit for bit given the GPU's own f32 spec,
one level-0 tile per image.  usage: python tests/fuzz_track_manager.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import thesia_amd as ta  # noqa: E402
from oracle import oracle as orc  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
cmap = open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
ctx = ta.Context(0)
t_end = time.time() + budget
n_ops = n_runs = 0
MAG_TOL = 1e-4
while time.time() < t_end:
    tm = ta.TrackManager(ctx)
    tm.set_colormap(cmap)
    tracks, next_id = {}, 0
    setting, dB_range = (40.0, 4, 1, ta.MEL), 100.0
    for _ in range(int(rng.integers(3, 9))):
        op = rng.choice(["add", "add", "remove", "setting", "range"])
        if op == "add" or not tracks:
            new = []
            for _ in range(int(rng.integers(1, 3))):
                sr = int(rng.choice([8000, 16000, 22050, 44100, 48000]))
                n = int(rng.integers(200, 30000))
                wav = (rng.standard_normal((int(rng.integers(1, 3)), n)) * 0.1).astype(np.float32)
                new.append((next_id, sr, wav))
                tracks[next_id] = (sr, wav)
                next_id += 1
            tm.add_tracks(new)
            tm.apply_track_list_changes()
        elif op == "remove":
            tid = int(rng.choice(list(tracks)))
            tm.remove_track(tid)
            del tracks[tid]
            tm.apply_track_list_changes()
        elif op == "setting":
            setting = (float(rng.choice([20.0, 40.0, 2048 / 48])), int(rng.choice([2, 4, 8])), int(rng.choice([1, 2])),
                       int(rng.choice([ta.MEL, ta.LINEAR])))
            tm.set_setting(*setting)
        else:
            dB_range = float(rng.choice([40.0, 60.0, 100.0, 120.0]))
            tm.set_dB_range(dB_range)
        n_ops += 1
        if not tracks:
            continue
        # ---- the oracle's view of the state
        max_sr = max(sr for sr, _ in tracks.values())
        mins, maxs, specs = [], [], {}
        for tid, (sr, wav) in tracks.items():
            hop, win, n_fft = orc.calc_framing_params(setting[0], setting[1], setting[2], sr)
            fb = orc.calc_mel_fb_default(sr, n_fft) if setting[3] == ta.MEL else None
            for ch in range(wav.shape[0]):
                want = orc.calc_spec(wav[ch], win, hop, n_fft, mel_fb=fb)
                got = tm.spec(tid, ch)
                assert got.shape == want.shape, (setting, sr, got.shape, want.shape)
                a, b = np.power(10.0, got.astype(np.float64) / 20), np.power(10.0, want.astype(np.float64) / 20)
                scale = np.maximum(b.max(axis=1, keepdims=True), 1e-30)
                assert (np.abs(a - b) / scale).max() <= MAG_TOL, (setting, sr)
                specs[(tid, ch)] = (got, sr)
                mins.append(got.min())
                maxs.append(got.max())
        lo, hi = orc.global_db_range(mins, maxs, dB_range)
        glo, ghi, gsr = tm.db_state()
        assert (glo, ghi, gsr) == (lo, hi, max_sr), ((glo, ghi, gsr), (lo, hi, max_sr))
        for (tid, ch), (got, sr) in specs.items():
            r = orc.hz_range_to_idx(orc.MEL if setting[3] == ta.MEL else orc.LINEAR, (0.0, max_sr / 2), sr, got.shape[1])
            img = tm.img(tid, ch)
            assert np.array_equal(img, orc.convert_spectrogram_to_img(got, r, (lo, hi), 258)), (setting, sr, r)
            _, srev = tm.revisions()
            assert tm.get_spectrogram_tile(tid, ch, 0, 0, 0, 0) == orc.encode_spectrogram_tile(img, cmap, srev, 0, 0, 0, 0)
    tm.close()
    n_runs += 1
print(f"{n_runs} random sessions, {n_ops} operations: TrackManager state equal to the oracle pipeline after every step")
