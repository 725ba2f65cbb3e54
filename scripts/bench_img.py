#!/usr/bin/env python3
"""Image-stage microbenchmark (development tool): spec->u16 quantise/transpose and level-0 raster
on the bench.py workload shapes.  usage: python scripts/bench_img.py [--tracks N] [--frames T] [--reps R]"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import thesia_amd as ta  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--tracks", type=int, default=128)
ap.add_argument("--frames", type=int, default=2813)
ap.add_argument("--height", type=int, default=1025)
ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--sustain", type=int, default=0, help="also time N back-to-back launches (the second half of 2N) of the fused kernel and of th_dev_copy: "
                "the sustained rate — a launch between idle gaps runs on a cool, unthrottled part")
a = ap.parse_args()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
T, H, n = a.frames, a.height, a.tracks
ap2 = os.environ.get("TH_DENSE") == "1"
# TH_IMG_PAD_MB: allocate (and keep) that many MiB first — shifts where the buffers land (placement sensitivity)
_pad = torch.empty(int(float(os.environ.get("TH_IMG_PAD_MB", "0")) * (1 << 20)), dtype=torch.uint8, device=dev)
sp, ip = (H, T) if ap2 else (ta.pitch_f32(H), ta.pitch_u16(T))
# TH_IMG_DATA: uniform (default: every u16 level equally likely) | normal (noise floor around -60 dB) | steps (a smooth
# ramp along frequency + small noise: neighbouring pixels share LUT entries, like a real spectrogram)
kind = os.environ.get("TH_IMG_DATA", "uniform")
if kind == "uniform":
    spec = torch.rand((n, T, sp), device=dev) * -100.0
elif kind == "normal":
    spec = (torch.randn((n, T, sp), device=dev) * 12.0 - 60.0)
else:
    spec = (torch.linspace(-5.0, -95.0, sp, device=dev)[None, None, :] + torch.randn((n, T, sp), device=dev) * 3.0).contiguous()
d_rng = torch.tensor([-100.0, 0.0], dtype=torch.float32, device=dev)
use_drange = os.environ.get("TH_IMG_DRANGE") == "1"
img = torch.empty((n, H, ip), dtype=torch.int16, device=dev)
cmap = open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
d_cmap = torch.frombuffer(bytearray(cmap), dtype=torch.uint8).to(dev)
geoms = []
tx = 0
while True:
    ty, any_row = 0, False
    while True:
        g = ta.spectrogram_tile_geometry(T, H, 0, 0, tx, ty)
        if g.width == 0 or g.height == 0:
            break
        geoms.append(g); any_row = True; ty += 1
    if not any_row:
        break
    tx += 1
tile_px = sum(g.width * g.height for g in geoms)
tile_slots = sum(-(-(g.width * g.height) // 64) * 64 for g in geoms)  # every tile on a 256-byte boundary
rgba = torch.empty((n, tile_slots, 4), dtype=torch.uint8, device=dev)
imgd = (ta.ImgDesc * n)(*[ta.ImgDesc(spec[i].data_ptr(), img[i].data_ptr(), T, H, 0, H, sp, ip) for i in range(n)])
rast = []
for i in range(n):
    off = 0
    for g in geoms:
        rast.append(ta.RasterDesc(img[i].data_ptr(), rgba[i].data_ptr() + off * 4, T, H, g.origin_x, g.origin_y, g.width, g.height, ip, 0))
        off += -(-(g.width * g.height) // 64) * 64
rast = (ta.RasterDesc * len(rast))(*rast)


def timeit(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(a.reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


px = n * T * H
ms = timeit(lambda: ctx.spec_to_img_batch(imgd, -100.0, 0.0, 258))
print(f"spec_to_img: {ms:.3f} ms  {px * 6 / ms / 1e6:.0f} GB/s ({px * 6 / ms / 1e6 / 80:.1f}% of 8 TB/s)  {px / ms / 1e3:.0f} Mpx/s")
ms = timeit(lambda: ctx.raster_tiles(rast, d_cmap.data_ptr(), 258))
rpx = n * tile_px
print(f"raster_level0: {ms:.3f} ms  {rpx * 6 / ms / 1e6:.0f} GB/s ({rpx * 6 / ms / 1e6 / 80:.1f}% of 8 TB/s)  {rpx / ms / 1e3:.0f} Mpx/s")

# round 4: the fused pass (th_spec_to_img_raster_batch_dev): same outputs, 10 B per pixel
items = []
for i in range(n):
    ptrs, off = [], 0
    for g in geoms:
        ptrs.append(rgba[i].data_ptr() + off * 4)
        off += -(-(g.width * g.height) // 64) * 64
    items.append((imgd[i], ptrs))
fused = ctx.make_img_tiles_descs(items)
for _ in range(2):
    if use_drange:
        ms = timeit(lambda: ctx.spec_to_img_raster_batch(fused, d_cmap.data_ptr(), 258, d_range=d_rng.data_ptr()))
    else:
        ms = timeit(lambda: ctx.spec_to_img_raster_batch(fused, d_cmap.data_ptr(), 258, min_dB=-100.0, max_dB=0.0))
    print(f"fused quantise + raster [{kind}{', device range' if use_drange else ''}]: {ms:.3f} ms  {px * 10 / ms / 1e6:.0f} GB/s ({px * 10 / ms / 1e6 / 80:.1f}% of 8 TB/s)  {px / ms / 1e3:.0f} Mpx/s")
    ms2 = timeit(lambda: (ctx.spec_to_img_batch(imgd, -100.0, 0.0, 258), ctx.raster_tiles(rast, d_cmap.data_ptr(), 258)))
    print(f"the two kernels: {ms2:.3f} ms")

if a.sustain:
    def sustained(fn, nrep):
        for _ in range(nrep):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(nrep):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / nrep
    nb = 1 << 30
    ca = torch.empty(nb // 4, dtype=torch.float32, device=dev).normal_()
    cb = torch.empty_like(ca)
    for rnd in range(2):
        if use_drange:
            ms = sustained(lambda: ctx.spec_to_img_raster_batch(fused, d_cmap.data_ptr(), 258, d_range=d_rng.data_ptr()), a.sustain)
        else:
            ms = sustained(lambda: ctx.spec_to_img_raster_batch(fused, d_cmap.data_ptr(), 258, min_dB=-100.0, max_dB=0.0), a.sustain)
        print(f"sustained x{a.sustain}: fused quantise + raster {ms:.3f} ms  {px * 10 / ms / 1e6:.0f} GB/s ({px * 10 / ms / 1e6 / 80:.1f}% of 8 TB/s)")
        ms = sustained(lambda: (ctx.spec_to_img_batch(imgd, -100.0, 0.0, 258), ctx.raster_tiles(rast, d_cmap.data_ptr(), 258)), a.sustain)
        print(f"sustained x{a.sustain}: the two kernels {ms:.3f} ms")
        ms = sustained(lambda: ctx.dev_copy(cb.data_ptr(), ca.data_ptr(), nb), a.sustain)
        print(f"sustained x{a.sustain}: th_dev_copy 1 GiB -> 1 GiB {ms:.3f} ms  {2 * nb / ms / 1e6:.0f} GB/s")
        ms = timeit(lambda: ctx.dev_copy(cb.data_ptr(), ca.data_ptr(), nb))
        print(f"one launch between synchronisations: th_dev_copy {ms:.3f} ms  {2 * nb / ms / 1e6:.0f} GB/s")
