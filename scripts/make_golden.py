#!/usr/bin/env python3
"""Generates the committed fixtures under tests/golden/.  Run in the build container only
(reads /root/reference for the colormap DATA table; the GPU box never runs this).

  colormap_inferno_rgba258.bin  the 258-entry RGBA8 colormap the frontend hands to init()
                                (src/prototypes/constants/colors.ts:65-165: black, 256 x inferno,
                                white; converted exactly as colors.ts:156-163 does)
  stft_f64_cases.npz            seeded inputs + float64 numpy.fft.rfft ground truth (|X| and dB) for
                                small framing cases — the mathematical DFT that stands in for the
                                un-vendored realfft/rustfft
"""
import os
import re
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden")


def colormap():
    src = open("/root/reference/src/prototypes/constants/colors.ts").read()
    body = src[src.index("const COLORMAP_RGBF = ["):src.index("]; // prettier-ignore")]
    rows = re.findall(r"\[\s*([0-9.]+),\s*([0-9.]+),\s*([0-9.]+)\s*\]", body)
    assert len(rows) == 258, len(rows)
    out = np.zeros((258, 4), np.uint8)
    for i, (r, g, b) in enumerate(rows):
        for j, v in enumerate((r, g, b)):
            out[i, j] = min(max(int(np.floor(float(v) * 255 + 0.5)), 0), 255)  # Math.round
        out[i, 3] = 255
    out.tofile(os.path.join(OUT, "colormap_inferno_rgba258.bin"))


def reflect_index(i, n):
    P = 2 * (n - 1)
    j = np.mod(i, P)
    return np.where(j < n, j, P - j)


def stft_cases():
    from tests.synth import synth_track
    cases = {}
    for name, (sr, win, hop, n_fft, n) in {
        "c1024": (48000, 1024, 256, 1024, 6000), "c2048": (48000, 2048, 512, 2048, 9000),
        "cdef48": (48000, 1920, 480, 2048, 8000), "cdef44": (44100, 1764, 441, 2048, 8000),
        "c4096": (48000, 4096, 1024, 4096, 12000), "cshort": (48000, 2048, 512, 2048, 700),
        "c64": (8000, 64, 16, 64, 500),
    }.items():
        x = synth_track(len(cases), sr, n)
        i = np.arange(win, dtype=np.float32)
        w = ((np.float32(0.5) - np.float32(0.5) * np.cos(np.float32(2) * (np.float32(np.pi) * i / np.float32(win)),
                                                          dtype=np.float32)) / np.float32(n_fft)).astype(np.float32)
        T = (n + 2 * (win // 2) - win) // hop + 1
        idx = (np.arange(T)[:, None] * hop - win // 2) + np.arange(win)[None, :]
        fr = (x[reflect_index(idx, n)] * w[None, :]).astype(np.float32)
        buf = np.zeros((T, n_fft))
        pl = (n_fft - win) // 2
        buf[:, pl:pl + win] = fr
        amp = np.abs(np.fft.rfft(buf, axis=1))
        cases[name + "_x"] = x
        cases[name + "_amp"] = amp.astype(np.float32)
        cases[name + "_par"] = np.array([sr, win, hop, n_fft], np.int64)
    np.savez_compressed(os.path.join(OUT, "stft_f64_cases.npz"), **cases)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    colormap()
    stft_cases()
    print(sorted(os.listdir(OUT)), sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes")
