#!/usr/bin/env python3
"""Generates tests/golden/lod_pillow_cases.npz: third-party known answers for LOD > 0 tiles.

The reference resamples with fast_image_resize 6.0.0 (Lanczos3, U16 pixels; render_tiles.rs:354-393).  That crate is not
vendored and cannot be built here, so LOD > 0 stays formally unpinned against it — but its convolution is the one Pillow's
`ImagingResample` defines (the crate documents itself as following it: precompute_coeffs with support = 3 * scale, window
[int(center - support + 0.5), int(center + support + 0.5)) clipped at the IMAGE, taps normalised by their sum, horizontal
pass then vertical pass with one rounding each, crop box in source coordinates).  Pillow IS in the build container, so its
16-bit path (mode "I;16", `Image.resize(size, LANCZOS, box=...)`) is used here as an independent implementation:

    python scripts/make_golden_lod.py          (build container only — Pillow never travels to the GPU box)

Inputs are closed-form integer images (tests/lod_images.py: no RNG, nothing to store); the fixture holds Pillow's outputs
for (a) whole-image resizes to the LOD dimensions of render_tiles.rs:290-313 and (b) `box=` crops with the tile geometry and
crop box of render_tiles.rs:382-386 — full pixels for the small cases and a few tiles of the large one, SHA-256 + a strided
sample for every other tile.  Only data leaves this script (a fixture is inputs and expected outputs).
"""
from __future__ import annotations

import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests.lod_images import IMAGES, LEVELS, TILE_IMAGE, TILE_LEVELS, FULL_TILES, lod_image, tile_geometry  # noqa: E402


def pillow_resize(img: np.ndarray, out_w: int, out_h: int, box=None) -> np.ndarray:
    from PIL import Image
    im = Image.fromarray(np.ascontiguousarray(img, dtype=np.uint16))
    assert im.mode == "I;16", im.mode
    r = im.resize((out_w, out_h), Image.LANCZOS, box=box)
    out = np.asarray(r, dtype=np.uint16)
    assert out.shape == (out_h, out_w)
    return out


def main() -> None:
    import PIL
    data = {"pillow_version": np.array(PIL.__version__)}
    for name in IMAGES:
        img = lod_image(name)
        Hh, W = img.shape
        for lx, ly in LEVELS[name]:
            w, h = -(-W // (1 << lx)), -(-Hh // (1 << ly))
            data[f"whole/{name}/{lx}_{ly}"] = pillow_resize(img, w, h)
    img = lod_image(TILE_IMAGE)
    Hh, W = img.shape
    for lx, ly in TILE_LEVELS:
        lod_w, lod_h = -(-W // (1 << lx)), -(-Hh // (1 << ly))
        for ty in range(-(-lod_h // 512)):
            for tx in range(-(-lod_w // 512)):
                g = tile_geometry(W, Hh, lx, ly, tx, ty)
                # crop box in source coordinates (render_tiles.rs:382-386), f64
                left, right = g["origin_x"] * W / lod_w, (g["origin_x"] + g["width"]) * W / lod_w
                top, bottom = g["origin_y"] * Hh / lod_h, (g["origin_y"] + g["height"]) * Hh / lod_h
                t = pillow_resize(img, g["width"], g["height"], box=(left, top, right, bottom))
                key = f"tile/{lx}_{ly}/{tx}_{ty}"
                data[key + "/sha256"] = np.frombuffer(hashlib.sha256(t.tobytes()).digest(), np.uint8)
                data[key + "/sample"] = t[::7, ::7].copy()
                if (lx, ly, tx, ty) in FULL_TILES:
                    data[key + "/full"] = t
    out = os.path.join(ROOT, "tests", "golden", "lod_pillow_cases.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out), "bytes,", len(data), "arrays")


if __name__ == "__main__":
    main()
