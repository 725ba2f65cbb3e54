#!/usr/bin/env python3
"""Development tool: which kernel route every UI-reachable Mel setting gets (sample rate x window ms x t_overlap x f_overlap).
usage: python scripts/ui_shapes.py [linear]"""
import os
import sys
from collections import Counter

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import thesia_amd as ta  # noqa: E402

scale = ta.LINEAR if len(sys.argv) > 1 and sys.argv[1] == "linear" else ta.MEL
ctx = ta.Context(0)
rows = Counter()
for sr in (8000, 11025, 16000, 22050, 24000, 32000, 44100, 48000, 88200, 96000, 176400, 192000):
    for win_ms in (10.0, 20.0, 40.0, 85.0, 170.0, 340.0):
        for t_ov in (2, 4, 8, 16):
            for f_ov in (1, 2, 4):
                hop, win, n_fft = ta.calc_framing_params(win_ms, t_ov, f_ov, sr)
                if n_fft > 32768 or n_fft < 256:
                    continue
                try:
                    p = ta.Plan(ctx, sr, win, hop, n_fft, scale, 0)
                except ta.ThError as e:
                    rows[(n_fft, "refused")] += 1
                    continue
                name = p.kernel_name
                rows[(n_fft, name)] += 1
                if ("+mel" in name or "generic" in name) and scale == ta.MEL:
                    print(f"sr {sr:6d} win {win_ms:5.0f} ms t_ov {t_ov:2d} f_ov {f_ov}: {win}/{hop}/{n_fft} {p.height} mels -> {name}  (moment table: {p.mel_moments_info()['groups']} groups, max_dev {p.mel_moments_info()['max_dev']:.1e})")
                p.close()
print()
for (n_fft, name), c in sorted(rows.items()):
    print(f"n_fft {n_fft:6d}: {c:3d} x {name}")
