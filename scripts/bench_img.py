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
a = ap.parse_args()
dev = torch.device("cuda", 0)
side = torch.cuda.Stream(dev)
torch.cuda.set_stream(side)
ctx = ta.Context(0, side.cuda_stream)
T, H, n = a.frames, a.height, a.tracks
ap2 = os.environ.get("TH_DENSE") == "1"
sp, ip = (H, T) if ap2 else (ta.pitch_f32(H), ta.pitch_u16(T))
spec = torch.rand((n, T, sp), device=dev) * -100.0
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
    ms = timeit(lambda: ctx.spec_to_img_raster_batch(fused, d_cmap.data_ptr(), 258, min_dB=-100.0, max_dB=0.0))
    print(f"fused quantise + raster: {ms:.3f} ms  {px * 10 / ms / 1e6:.0f} GB/s ({px * 10 / ms / 1e6 / 80:.1f}% of 8 TB/s)  {px / ms / 1e3:.0f} Mpx/s")
    ms2 = timeit(lambda: (ctx.spec_to_img_batch(imgd, -100.0, 0.0, 258), ctx.raster_tiles(rast, d_cmap.data_ptr(), 258)))
    print(f"the two kernels: {ms2:.3f} ms")
