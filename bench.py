#!/usr/bin/env python3
"""bench.py — headline benchmark of the spectrogram hot path (BASELINE.json metric:
"STFT frames/sec + spectrogram Mpixels/sec at n_fft=2048, 1/2/4/8 MI355X").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (the driver's form for N > 1)

`--gpus N` with N > 1 and no WORLD_SIZE in the environment makes this process a LAUNCHER: before anything touches the GPU
(no torch import, no HIP call) it checks that N devices are visible — fewer is an error, never a silent one-rank run —
and starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py <same arguments>` as a CHILD process
(never an exec), relays its stdout and exits with its code.  Under torchrun, WORLD_SIZE must equal --gpus.

TH_BENCH_SHARE_GPU=1 (a REHEARSAL, not a measurement of scaling): the N ranks all use GPU 0 and torch.distributed runs over
gloo (RCCL refuses two ranks on one device), so that the whole N > 1 control flow — launcher, sharding, the range exchange
between the two kernels of the step, max-over-ranks timing, the tile gather — runs with real processes on a one-GPU box.
The record says so (`config.ranks_share_one_gpu`), and its frames/s is what N interleaved jobs get from ONE card.

Workload (per GPU, fixed as N grows -> weak scaling): BASELINE config "1024 synthetic 48 kHz mono
tracks, n_fft=2048, sharded across 8 GPUs" = 128 tracks x 30 s per GPU (track index = rank*128+i),
Hann win 2048 / hop 512, linear-frequency dB.  At N = 8 this is exactly that config; at N = 1 it
is its one-GPU shard (the single 60 s track of config[1] is 35 MB of traffic — a ~10 us launch that
cannot load 256 CUs; it is reported beside the headline as `single_track_cfg2`).

One step = one pass of the hot path over the resident batch:
  STFT -> |X| -> 20 log10 (+ fused per-track min/max)          th_calc_spec_batch_dev
  global dB range (2-float all-reduce over ranks when N > 1)    core/mod.rs:169-180
  f32 dB -> u16 grey image, transposed                          th_spec_to_img_batch_dev
  level-0 colormap raster of every tile -> RGBA                 th_raster_tiles_dev
Inputs are resident in HBM when the timed region starts.  value = frames of all ranks / time.
PyTorch provides device memory, the stream and torch.distributed (RCCL); all compute is the
library's own HIP kernels through the C ABI.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SHARE_GPU = os.environ.get("TH_BENCH_SHARE_GPU") == "1"   # N ranks on GPU 0 over gloo: a control-flow rehearsal (module docstring)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s measured copy
MFMA_F32_PEAK_TFLOPS = 157.3  # dense fp32 MFMA peak (MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32)


def dist_of(ms):
    """min / median / p90 / max of a series of launch durations (ms)"""
    a = np.sort(np.asarray(ms, dtype=np.float64))
    if a.size == 0:
        return None
    return {"n": int(a.size), "min": float(a[0]), "median": float(np.median(a)), "p90": float(a[min(a.size - 1, int(np.ceil(0.9 * a.size)) - 1)]),
            "max": float(a[-1]), "mean": float(a.mean())}


def memory_skeleton():
    """The STFT kernel's memory skeleton, timed on this box in this run (scripts/ubench/stft_skeleton.hip, built by
    __graft_entry__.build()): the same persistent waves, chunk queue, hop loads a frame ahead and row stores as
    stft_wave_kernel on the same workload shape, with NO arithmetic — what the memory system gives this access structure —
    and with the kernel's amount of stand-in VALU / LDS work.  A child process (its own 2.3 GB of buffers)."""
    import subprocess
    exe = os.path.join(ROOT, "scripts", "ubench", "stft_skeleton")
    if not os.path.exists(exe):
        return {"error": "scripts/ubench/stft_skeleton not built"}
    out = {"source": "scripts/ubench/stft_skeleton.hip (mode 7), child process, HIP events, median of 20 launches"}
    for key, gap in (("gaps_1ms", "1000"), ("back_to_back", "0")):
        try:
            r = subprocess.run([exe, gap, "7"], capture_output=True, text=True, timeout=120)
            j = json.loads(r.stdout.strip().splitlines()[-1])
            out[key] = {"no_work_ms": j["runs"][0]["median_ms"], "no_work_min_ms": j["runs"][0]["min_ms"],
                        "with_kernel_amount_of_work_ms": j["runs"][1]["median_ms"]}
        except Exception as e:  # report, do not fail the bench line
            out[key] = {"error": str(e)[:200]}
    return out


def visible_gpu_count(env=None) -> int:
    """GPUs a rank could open, found WITHOUT initialising HIP in this process: the KFD topology (nodes with SIMDs are
    GPUs), cut down by HIP_/ROCR_/CUDA_VISIBLE_DEVICES.  Where the topology cannot be read, a CHILD process asks torch
    (torch.cuda.device_count() does not create a HIP context on this image; a child keeps even that out of the launcher).
    TH_BENCH_ASSUME_GPUS overrides (tests)."""
    env = os.environ if env is None else env
    if env.get("TH_BENCH_ASSUME_GPUS", "") != "":
        return int(env["TH_BENCH_ASSUME_GPUS"])
    n = None
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        n = 0
        for node in os.listdir(base):
            props = dict(ln.split()[:2] for ln in open(os.path.join(base, node, "properties")) if len(ln.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except Exception:
        n = None
    if n is None:
        import subprocess
        try:
            r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True,
                               text=True, timeout=600, env=dict(env))
            return int(r.stdout.strip().splitlines()[-1])
        except Exception:
            return 0
    # a container may see more GPUs in /sys than its device cgroup lets it open: a GPU counts only with a render node this
    # process can open (over-counting surfaced only as a rank's late failure, the other ranks stuck in rendezvous)
    try:
        usable = 0
        for d in os.listdir("/dev/dri"):
            if d.startswith("renderD"):
                try:   # open + close of a render node: what the cgroup decides, and no HIP context
                    os.close(os.open(os.path.join("/dev/dri", d), os.O_RDWR))
                    usable += 1
                except OSError:
                    pass
        n = min(n, usable)
    except Exception:
        pass
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = env.get(var)
        if v is not None:
            n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    return n


def launcher_plan(gpus: int, argv, env, n_visible: int):
    """What `bench.py --gpus N` (N > 1, no WORLD_SIZE) starts: (command, environment) of ONE child process — torchrun with
    N ranks on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve), this script and its own
    arguments unchanged.  Fewer than N visible devices: SystemExit with a message, never a smaller job."""
    share = env.get("TH_BENCH_SHARE_GPU") == "1"   # rehearsal: every rank on GPU 0, gloo (see the module docstring)
    if share and n_visible >= 1:
        n_visible = gpus
    if n_visible < gpus:
        raise SystemExit(f"bench.py: --gpus {gpus} needs {gpus} visible GPUs, this node shows {n_visible}; refusing to run a smaller "
                         "job under the same name (set HIP_VISIBLE_DEVICES / pick --gpus to match)")
    # rendezvous on 127.0.0.1.  A port the environment names (the driver's form) is used as given; otherwise torchrun's own
    # c10d rendezvous takes a free port itself (endpoint port 0) — a port picked HERE by bind-and-close could be taken by
    # another process before torchrun binds it
    port = env.get("MASTER_PORT")
    if port:
        rdzv = ["--master-addr", "127.0.0.1", "--master-port", str(port)]
    else:
        rdzv = ["--rdzv-backend=c10d", "--rdzv-endpoint=127.0.0.1:0", "--local-addr", "127.0.0.1"]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}"] + rdzv + [os.path.abspath(__file__)] + list(argv)
    child_env = dict(env)
    child_env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    child_env.setdefault("OMP_NUM_THREADS", "4")
    return cmd, child_env


def launch_ranks(args, argv) -> int:
    """The launcher side of `--gpus N`: start the ranks as a child, pass its stdout through line by line (the JSON record
    stays the last line), return its exit code.  Nothing here imports torch or touches the GPU."""
    import subprocess
    cmd, child_env = launcher_plan(args.gpus, argv, os.environ, visible_gpu_count())
    print("bench.py launcher: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=child_env, stdout=subprocess.PIPE, text=True, bufsize=1, cwd=ROOT)
    try:
        for line in proc.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return proc.wait()
    except BaseException:
        proc.terminate()   # the exact child we started (and through torchrun its ranks), nothing matched by pattern
        try:
            proc.wait(timeout=30)
        except Exception:
            proc.kill()
        raise


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tracks-per-gpu", type=int, default=128)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--spin-up-steps", type=int, default=64, help="untimed steps before the warm-up steps (clock settling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-skeleton", action="store_true", help="skip the memory-skeleton child process (roofline.memory_skeleton)")
    ap.add_argument("--no-single-track", action="store_true", help="skip the extras (single track, other framings, tile latency, ...)")
    ap.add_argument("--no-full-cfg5", action="store_true", help="skip the strong-scaling anchor (all 1024 tracks of config 5 on one GPU)")
    ap.add_argument("--kernel", type=int, default=0, help="0 auto, 1 generic, 2 wave")
    ap.add_argument("--no-fused-image", action="store_true", help="quantise and raster as two kernels (A/B; the default fuses them)")
    return ap.parse_args(argv)


def synth_on_gpu(torch, dev, track_ids, sr: int, n: int):
    """tests/synth.py's signal, evaluated on the GPU (f64 phase, same seeded parameters)."""
    from tests.synth import track_params
    out = torch.empty((len(track_ids), n), dtype=torch.float32, device=dev)
    t = torch.arange(n, dtype=torch.float64, device=dev) / sr
    dur = n / sr
    for i, tid in enumerate(track_ids):
        freqs, amps, phases, chirp_amp, noise_seed = track_params(tid, sr)
        x = torch.zeros(n, dtype=torch.float64, device=dev)
        for f, a, p in zip(freqs, amps, phases):
            x += a * torch.sin(2 * np.pi * f * t + p)
        x += chirp_amp * torch.sin(2 * np.pi * (100.0 * t + 0.5 * (0.4 * sr - 100.0) / dur * t * t))
        g = torch.Generator(device=dev)
        g.manual_seed(noise_seed)
        x += (torch.rand(n, dtype=torch.float64, device=dev, generator=g) * 2 - 1) * 1e-3
        out[i] = x.clamp_(-1.0, 1.0).to(torch.float32)
    return out


class Workload:
    """Device-resident batch + descriptor tables for one GPU."""

    def __init__(self, torch, ta, ctx, dev, track_ids, sr, n, win, hop, n_fft, kernel, cmap_bytes, base=None, scale=None, n_mel=0):
        self.torch, self.ta, self.ctx = torch, ta, ctx
        n_tracks = len(track_ids)
        self.n_tracks, self.n = n_tracks, n
        self.plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.LINEAR if scale is None else scale, n_mel)
        if kernel:
            self.plan.set_kernel(kernel)
        self.T, self.H = self.plan.n_frames(n), self.plan.height
        # base: another Workload whose audio is reused round-robin (the 1024-track strong-scaling extra: synthesising 1024
        # tracks would take longer than the whole bench; every track still gets its own spec, image and tiles)
        self.wav = synth_on_gpu(torch, dev, track_ids, sr, n) if base is None else base.wav
        wav_of = (lambda i: self.wav[i]) if base is None else (lambda i: self.wav[i % base.n_tracks])
        # HBM layout: rows padded to 128 B (th_pitch_*); the dense reference layout is what copy-out returns
        self.sp, self.ip = ta.pitch_f32(self.H), ta.pitch_u16(self.T)
        self.spec = torch.empty((n_tracks, self.T, self.sp), dtype=torch.float32, device=dev)
        self.img = torch.empty((n_tracks, self.H, self.ip), dtype=torch.int16, device=dev)
        self.minmax = torch.empty((n_tracks, 2), dtype=torch.float32, device=dev)
        self.cmap = torch.frombuffer(bytearray(cmap_bytes), dtype=torch.uint8).to(dev)
        self.n_colors = len(cmap_bytes) // 4
        # every level-0 tile of every image (render_tiles.rs:290-313 geometry, gutters included)
        geoms = []
        tx = 0
        while True:
            ty, any_row = 0, False
            while True:
                g = ta.spectrogram_tile_geometry(self.T, self.H, 0, 0, tx, ty)
                if g.width == 0 or g.height == 0:
                    break
                geoms.append(g)
                any_row = True
                ty += 1
            if not any_row:
                break
            tx += 1
        self.tile_px = sum(g.width * g.height for g in geoms)
        # every tile starts on a 256-byte boundary (the rasteriser stores 16 bytes per lane: an unaligned tile base
        # degrades it to 4-byte stores)
        self.tile_slots = sum(-(-(g.width * g.height) // 64) * 64 for g in geoms)
        self.rgba = torch.empty((n_tracks, self.tile_slots, 4), dtype=torch.uint8, device=dev)
        self.chan = (ta.ChanDesc * n_tracks)(*[
            ta.ChanDesc(wav_of(i).data_ptr(), self.spec[i].data_ptr(), n, self.T, self.sp) for i in range(n_tracks)])
        self.imgd = (ta.ImgDesc * n_tracks)(*[
            ta.ImgDesc(self.spec[i].data_ptr(), self.img[i].data_ptr(), self.T, self.H, 0, self.H, self.sp, self.ip)
            for i in range(n_tracks)])
        rast = []
        for i in range(n_tracks):
            off = 0
            for g in geoms:
                rast.append(ta.RasterDesc(self.img[i].data_ptr(), self.rgba[i].data_ptr() + off * 4, self.T,
                                               self.H, g.origin_x, g.origin_y, g.width, g.height, self.ip, 0))
                off += -(-(g.width * g.height) // 64) * 64
        self.rast = (ta.RasterDesc * len(rast))(*rast)
        # the same outputs through ONE pass over the spec (th_spec_to_img_raster_batch_dev, round 4): per image the level-0
        # tile grid's device pointers, tile (tx, ty) at tx * n_ty + ty — the order `geoms` was built in
        fused_items = []
        for i in range(n_tracks):
            ptrs, off = [], 0
            for g in geoms:
                ptrs.append(self.rgba[i].data_ptr() + off * 4)
                off += -(-(g.width * g.height) // 64) * 64
            fused_items.append((self.imgd[i], ptrs))
        self.fused = ctx.make_img_tiles_descs(fused_items)
        self.use_fused = True
        self.frames = n_tracks * self.T
        self.pixels = n_tracks * self.H * self.T
        self.range2 = torch.empty(2, dtype=torch.float32, device=dev)    # [min, -max]
        self.range_db = torch.empty(2, dtype=torch.float32, device=dev)  # [min_dB, max_dB]
        self.ev = []
        self.plan.time_kernel(True)  # HIP events around the dominant kernel launch, on the launch stream

    def step(self, dist=None, record=False):
        torch, ta = self.torch, self.ta
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(5)] if record else None
        if record:
            ev[0].record()
        if dist is None:  # one GPU: STFT + the global dB range in one call (th_calc_spec_batch_ranged_dev)
            self.plan.calc_spec_batch_ranged_dev(self.chan, self.minmax.data_ptr(), 100.0, self.range_db.data_ptr())
        else:
            self.plan.calc_spec_batch_dev(self.chan, self.minmax.data_ptr())
        if record:
            ev[1].record()
        # global dB range over every resident spec of every rank (core/mod.rs:169-180), without leaving the device:
        # [min, -max] of this rank -> (N > 1: the path's only exchange step, a 2-float MIN all-reduce) -> clamp
        if dist is not None:
            self.ctx.minmax_reduce_dev(self.minmax.data_ptr(), self.n_tracks, self.range2.data_ptr())
            if SHARE_GPU:   # gloo rehearsal on one card: through host memory (a synchronisation inside the step: not a timing to quote)
                r2 = self.range2.cpu()
                dist.all_reduce(r2, op=dist.ReduceOp.MIN)
                self.range2.copy_(r2)
            else:
                dist.all_reduce(self.range2, op=dist.ReduceOp.MIN)
            self.ctx.global_db_range_dev(self.range2.data_ptr(), 100.0, self.range_db.data_ptr())

        if record:
            ev[2].record()
        if self.use_fused:  # quantise + level-0 raster of every tile in one kernel (10 B per pixel)
            self.ctx.spec_to_img_raster_batch(self.fused, self.cmap.data_ptr(), self.n_colors, d_range=self.range_db.data_ptr())
            if record:
                ev[3].record()
        else:               # the two kernels (6 + 6 B per pixel)
            self.ctx.spec_to_img_batch_ranged(self.imgd, self.range_db.data_ptr(), 258)
            if record:
                ev[3].record()
            self.ctx.raster_tiles(self.rast, self.cmap.data_ptr(), self.n_colors)
        if record:
            ev[4].record()
            self.ev.append(ev)

    def stft_only(self):
        self.plan.calc_spec_batch_dev(self.chan, self.minmax.data_ptr())


def cpu_baseline(cmap_bytes, sr, n, win, hop, n_fft, target_s=15.0):
    """The oracle (CPU restatement of the reference algorithm, NOT rustfft) on this box's host
    cores: same step (STFT->dB->min/max->u16->level-0 RGBA) on a bounded sample of the workload,
    one task per track as the reference does when #channels >= #threads (core/mod.rs:152-163).
    Each task is ONE C call (oracle orc_track_step; ctypes releases the GIL)."""
    from concurrent.futures import ThreadPoolExecutor

    from oracle import oracle as orc
    from tests.synth import synth_track
    try:
        import psutil
        cores = psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        cores = os.cpu_count()
    cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    try:  # a cgroup CPU quota (containers) bounds the threads that can actually run
        q = open("/sys/fs/cgroup/cpu.max").read().split()
        if q[0] != "max":
            cores = max(1, min(cores, int(-(-int(q[0]) // int(q[1])))))
    except Exception:
        pass
    n_s = min(n, 10 * sr)  # 10 s tracks keep the sample bounded
    pool_tracks = [synth_track(i, sr, n_s) for i in range(8)]  # inputs generated outside the timed region

    def one(i):
        return orc.track_step(pool_tracks[i % len(pool_tracks)], win, hop, n_fft, cmap_bytes)

    t0 = time.perf_counter()
    frames1 = one(0)  # calibration (also warms the library)
    dt1 = time.perf_counter() - t0
    n_tasks = int(max(cores, min(64 * cores, round(target_s / max(dt1, 1e-3)) * cores)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, range(n_tasks)))
    wall = time.perf_counter() - t0
    frames = sum(res)
    return {"value": frames / wall, "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{n_tasks} tracks x {n_s / sr:.0f} s 48 kHz mono, n_fft={n_fft} hop={hop}, "
                      f"same step (STFT->dB->min/max->u16->level-0 RGBA), {wall:.1f} s wall; "
                      "CPU restatement of the reference algorithm (f32 radix-2 FFT), not rustfft"}


def tm_end_to_end_and_latency(torch, ta, ctx, wl, sr, cmap_bytes, n_tracks=32):
    """(1) PCIe-inclusive end to end through the C ABI's TrackManager mirror: host f32 tracks -> th_tm_add_tracks (upload,
    waveform pyramid, STFT) -> th_tm_apply_track_list_changes (dB range, u16 images, LOD mip pyramid) -> every level-0 tile
    of every track fetched to host memory.  (2) The tile-request latency path (lib.rs:342-389): per-request wall time of
    th_tm_get_spectrogram_tile / th_tm_get_waveform_tile at 1 and 8 reader threads, next to the oracle's CPU time for the
    same tiles."""
    import ctypes as C
    import threading

    from oracle import oracle as orc
    from thesia_amd import _ffi
    lib = _ffi.lib
    host = wl.wav[:n_tracks].cpu().numpy()
    n = host.shape[1]
    # The sequence twice, the SECOND one reported: a process's first add / apply also pays what no later one does (code objects of the
    # pyramid / quantise / mip kernels on first use, the runtime's pageable staging buffers, first-touch page faults of `host` — 20 ms
    # typically, 117 ms seen once); the first call's time is in the record as first_call_*.
    first_call = None
    for rep in range(2):
        tm = ta.TrackManager(ctx)
        tm.set_setting(2048 / 48, 4, 1, ta.LINEAR)
        tm.set_colormap(cmap_bytes)
        t0 = time.perf_counter()
        tm.add_tracks([(i, sr, host[i][None]) for i in range(n_tracks)])
        t1 = time.perf_counter()
        tm.apply_track_list_changes()
        t2 = time.perf_counter()
        if rep == 0:
            first_call = {"first_call_upload_pyramid_stft_ms": (t1 - t0) * 1e3, "first_call_range_quantise_mips_ms": (t2 - t1) * 1e3}
            tm.close()
    # the same tracks handed over in PINNED host memory (th_host_alloc): the upload then runs at PCIe speed
    pinned_in = None
    try:
        pin = C.c_void_p()
        _ffi.check(lib.th_host_alloc(ctx.handle, host.nbytes, C.byref(pin)))
        ph = np.ctypeslib.as_array((C.c_float * host.size).from_address(pin.value)).reshape(host.shape)
        ph[:] = host
        tm2 = ta.TrackManager(ctx)
        tm2.set_setting(2048 / 48, 4, 1, ta.LINEAR)
        tm2.set_colormap(cmap_bytes)
        tp0 = time.perf_counter()
        tm2.add_tracks([(i, sr, ph[i][None]) for i in range(n_tracks)])
        tp1 = time.perf_counter()
        tm2.close()
        del ph
        _ffi.check(lib.th_host_free(ctx.handle, pin))
        pinned_in = {"upload_pyramid_stft_ms": (tp1 - tp0) * 1e3, "host_input_GBs": host.nbytes / (tp1 - tp0) / 1e9}
    except Exception as e:  # extras must not break the bench line
        pinned_in = {"error": str(e)[:200]}
    ih, iw = tm.img(0, 0).shape
    tiles_xy = [(tx, ty) for tx in range(-(-iw // 512)) for ty in range(-(-ih // 512))]
    buf = np.empty(ta.api.SPECTROGRAM_TILE_MAX_BYTES, np.uint8)
    bp = buf.ctypes.data_as(_ffi.c_u8p)
    ln = C.c_size_t()
    t3 = time.perf_counter()
    nbytes = 0
    for i in range(n_tracks):
        for tx, ty in tiles_xy:
            _ffi.check(lib.th_tm_get_spectrogram_tile(tm.handle, i, 0, 0, 0, tx, ty, bp, buf.size, C.byref(ln)))
            nbytes += ln.value
    t4 = time.perf_counter()
    # the same tiles through th_tm_get_spectrogram_tiles: one launch, one transfer — into pinned memory the kernel writes
    # directly (th_host_alloc), and into a pageable buffer (pinned staging + host copy)
    batch = {}
    try:
        reqs_all = [(i, 0, 0, 0, tx, ty) for i in range(n_tracks) for tx, ty in tiles_xy]
        arr = (_ffi.TileRequest * len(reqs_all))(*[_ffi.TileRequest(*r, 0) for r in reqs_all])
        offs = (C.c_size_t * (len(reqs_all) + 1))()
        need = C.c_size_t()
        lib.th_tm_get_spectrogram_tiles(tm.handle, arr, len(reqs_all), None, 0, offs, C.byref(need))
        pin = C.c_void_p()
        _ffi.check(lib.th_host_alloc(ctx.handle, need.value, C.byref(pin)))
        page = np.empty(need.value, np.uint8)
        for key, ptr in (("pinned", pin), ("pageable", page.ctypes.data_as(C.c_void_p))):
            best = 1e9
            for _ in range(3):
                tb = time.perf_counter()
                _ffi.check(lib.th_tm_get_spectrogram_tiles(tm.handle, arr, len(reqs_all), ptr, need.value, offs, C.byref(need)))
                best = min(best, time.perf_counter() - tb)
            batch[key + "_ms"] = best * 1e3
            batch[key + "_GBs"] = need.value / best / 1e9
        batch["bytes"] = need.value
        batch["tiles"] = len(reqs_all)
        _ffi.check(lib.th_host_free(ctx.handle, pin))
    except Exception as e:  # extras must not break the bench line
        batch = {"error": str(e)[:200]}
    # the interactive case: the dB-range slider (lib.rs:257-266 -> core/mod.rs:123-126) re-quantises every image and
    # rebuilds every mip pyramid from the resident specs, no STFT
    t4b = time.perf_counter()
    tm.set_dB_range(80.0)
    t5 = time.perf_counter()
    tm.set_dB_range(100.0)
    t6 = time.perf_counter()
    frames = n_tracks * ta.stft_n_frames(n, 2048, 512)
    e2e = {"workload": f"{n_tracks} tracks x {n / sr:.0f} s 48 kHz mono from pageable host memory, n_fft=2048 hop=512: th_tm_add_tracks "
                       "-> th_tm_apply_track_list_changes -> every level-0 tile to host memory",
           "frames": frames, "upload_pyramid_stft_ms": (t1 - t0) * 1e3, "range_quantise_mips_ms": (t2 - t1) * 1e3, **(first_call or {}),
           "timed": "the second add_tracks / apply_track_list_changes of the process (a fresh TrackManager each time); first_call_* = the first",
           "all_level0_tiles_ms": (t4 - t3) * 1e3, "tile_bytes": nbytes, "all_level0_tiles_one_batch": batch,
           "set_dB_range_ms": min(t5 - t4b, t6 - t5) * 1e3,
           "frames_per_s_compute_only": frames / (t2 - t0), "frames_per_s_with_tile_fetch": frames / ((t2 - t0) + (t4 - t3)),
           "host_input_GBs": host.nbytes / (t1 - t0) / 1e9, "from_pinned_host_memory": pinned_in,
           "frames_per_s_with_batched_tile_fetch": (frames / ((t2 - t0) + batch["pinned_ms"] * 1e-3)) if "pinned_ms" in batch else None}

    rng = np.random.default_rng(1)
    n_wave_tiles = -(-n // 1024)

    def requests(kind, count):
        out = []
        for _ in range(count):
            i = int(rng.integers(0, n_tracks))
            if kind == "spec_level0":
                tx, ty = tiles_xy[int(rng.integers(0, len(tiles_xy)))]
                out.append(("s", i, 0, 0, tx, ty))
            elif kind == "spec_lod_1_0":
                out.append(("s", i, 1, 0, int(rng.integers(0, -(-(-(-iw // 2)) // 512))), int(rng.integers(0, -(-ih // 512)))))
            elif kind == "spec_lod_2_1":
                out.append(("s", i, 2, 1, int(rng.integers(0, -(-(-(-iw // 4)) // 512))), int(rng.integers(0, -(-(-(-ih // 2)) // 512)))))
            else:  # waveform tiles: a different tile every time (the LRU in front would otherwise answer)
                lvl = int(rng.integers(0, 6))
                out.append(("w", i, lvl, int(rng.integers(0, max(1, -(-n_wave_tiles // (1 << lvl)))))))
        return out

    def run(reqs, lat):
        b = np.empty(ta.api.SPECTROGRAM_TILE_MAX_BYTES, np.uint8)
        p = b.ctypes.data_as(_ffi.c_u8p)
        m = C.c_size_t()
        for r in reqs:
            t = time.perf_counter()
            if r[0] == "s":
                rc = lib.th_tm_get_spectrogram_tile(tm.handle, r[1], 0, r[2], r[3], r[4], r[5], p, b.size, C.byref(m))
            else:
                rc = lib.th_tm_get_waveform_tile(tm.handle, r[1], 0, r[2], r[3], p, b.size, C.byref(m))
            lat.append(time.perf_counter() - t)
            if rc != 0:
                raise RuntimeError(_ffi.lib.th_last_error())

    def measure(kind, threads, per_thread=120):
        cache = tm.tile_cache()
        cache.set_budget(1)  # every waveform request misses the LRU: the latency of the path behind it
        lats = [[] for _ in range(threads)]
        run(requests(kind, 10), [])  # warm-up (reader slots, mip tables)
        ths = [threading.Thread(target=run, args=(requests(kind, per_thread), lats[k])) for k in range(threads)]
        t = time.perf_counter()
        for th_ in ths:
            th_.start()
        for th_ in ths:
            th_.join()
        wall = time.perf_counter() - t
        a = np.sort(np.concatenate([np.asarray(x) for x in lats])) * 1e6
        return {"p50_us": float(a[len(a) // 2]), "p99_us": float(a[min(len(a) - 1, int(len(a) * 0.99))]),
                "requests_per_s": len(a) / wall}

    latency = {"what": "wall time per request at the C ABI (ctypes), one stream + one mapped pinned buffer per request (the raster kernel writes it directly); "
                       "LOD tiles are crops of the resident mip pyramid", "image": [int(ih), int(iw)]}
    for kind in ("spec_level0", "spec_lod_1_0", "spec_lod_2_1", "waveform"):
        latency[kind] = {"threads_1": measure(kind, 1), "threads_8": measure(kind, 8)}
    tm.set_lod_source(per_request=True)
    latency["spec_lod_1_0_per_request_resize"] = {"threads_1": measure("spec_lod_1_0", 1, 40)}
    tm.set_lod_source(per_request=False)
    # the oracle (CPU restatement, one core) on the same kinds of tile
    img = tm.img(0, 0)
    cpu = {}
    for kind, args_ in (("spec_level0", (0, 0, 1, 0)), ("spec_lod_1_0", (1, 0, 0, 0)), ("spec_lod_2_1", (2, 1, 0, 0))):
        t = time.perf_counter()
        for _ in range(3):
            orc.encode_spectrogram_tile(img, cmap_bytes, 1, *args_)
        cpu[kind + "_us"] = (time.perf_counter() - t) / 3 * 1e6
    t = time.perf_counter()
    for k in range(20):
        orc.encode_waveform_tile(host[0], 1, 3, k)
    cpu["waveform_level3_us"] = (time.perf_counter() - t) / 20 * 1e6
    latency["cpu_oracle_one_core"] = cpu
    tm.close()
    return e2e, latency


def main():
    args = parse_args()
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ:
        # launcher: N ranks as a child process, before anything here touches the GPU (TH_BENCH_FORCE_LAUNCHER=1 takes the
        # same route with N = 1: launcher -> torchrun -> rank -> RCCL process group, end to end on a one-GPU box)
        if args.gpus > 1 or os.environ.get("TH_BENCH_FORCE_LAUNCHER") == "1":
            sys.exit(launch_ranks(args, sys.argv[1:]))
    elif int(os.environ["WORLD_SIZE"]) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={os.environ['WORLD_SIZE']}: the launcher's rank count and "
                         "--gpus must agree (n_gpus in the record is the number of ranks that ran)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if rank != 0:  # only rank 0 reports; keep library banners (RCCL prints one through C stdio) of the others off stdout
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)

    import __graft_entry__ as ge
    ge.build()  # no-op when up to date; serialised across ranks by a file lock
    import torch
    import thesia_amd as ta

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if SHARE_GPU:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit(f"bench.py: rank {rank} wants device {local_rank}, only {torch.cuda.device_count()} visible")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    # under torchrun (WORLD_SIZE set) the process group always comes up, also at one rank; the env switch does the same
    # for a plain `python bench.py`: both exercise the RCCL path on one GPU
    if world > 1 or "WORLD_SIZE" in os.environ or os.environ.get("TH_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if SHARE_GPU:
            dist_mod.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        dist = dist_mod

    sr, (hop, win, n_fft) = 48000, ta.calc_framing_params(2048 / 48, 4, 1, 48000)
    n = int(round(args.seconds * sr))
    cmap_bytes = open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()

    # one side stream shared by torch (plumbing ops, events, RCCL) and the library's kernels
    side = torch.cuda.Stream(dev)
    torch.cuda.set_stream(side)
    ctx = ta.Context(local_rank, side.cuda_stream)
    # shard the global track list over ranks by frame count (th_shard_assign; equal tracks -> round-robin)
    total_tracks = args.tracks_per_gpu * world
    owner = ta.shard_assign([ta.stft_n_frames(n, win, hop)] * total_tracks, world)
    mine = [i for i in range(total_tracks) if owner[i] == rank]
    wl = Workload(torch, ta, ctx, dev, mine, sr, n, win, hop, n_fft, args.kernel, cmap_bytes)
    wl.use_fused = not args.no_fused_image

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # Spin-up (untimed, before the W warm-up steps, reported as `spin_up_steps`): a GPU that sat idle while the host built
    # the descriptor tables needs ~30-40 ms of sustained load before its power management has settled the clocks — round 2's
    # driver run (5 warm-up steps = 7 ms) timed exactly that transient: the STFT launch went 0.64 -> 0.48 ms monotonically
    # over its 20 timed steps (`roofline.launch_ms_series` shows the series of THIS run).  The timed region is untouched:
    # still exactly K steps between two barriers.
    for _ in range(args.spin_up_steps):
        wl.step(dist)
    for _ in range(args.warmup):
        wl.step(dist)
    barrier()
    wl.plan.time_kernel(True)  # reset the launch-duration history: only the timed region is kept
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wl.step(dist, record=True)
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cpu" if SHARE_GPU else dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # every rank must have quantised against the SAME global dB range (the path's one exchange): collect the ranks' [min_dB, max_dB]
    range_agrees = None
    if dist is not None:
        mine_r = wl.range_db.detach().cpu() if SHARE_GPU else wl.range_db.detach().clone()
        all_r = [torch.empty_like(mine_r) for _ in range(world)]
        dist.all_gather(all_r, mine_r)
        range_agrees = bool(all(torch.equal(all_r[0], r) for r in all_r))
        global_range = [float(v) for v in all_r[0].cpu()]
    # HIP events on the launch stream, inside the timed region: per-kernel average launch durations
    stft_stage_ms = float(np.mean([e[0].elapsed_time(e[1]) for e in wl.ev]))  # init + STFT kernel + boundary-frame kernel
    stft_series = [float(v) for v in wl.plan.kernel_ms_history()[-args.steps:]]
    stft_ms = float(np.mean(stft_series))                                        # the dominant kernel launch alone
    # image stage inside the timed step: one fused kernel (default) or the two kernels
    image_ms = float(np.mean([e[2].elapsed_time(e[4]) for e in wl.ev]))
    # the path's only exchange step, as it sits between the two kernels of the timed step (N > 1 or TH_BENCH_FORCE_DIST):
    # fold of the per-track (min, max) -> 2-float all-reduce (RCCL) -> clamp kernel; HIP events on the launch stream
    range_exchange_ms = float(np.mean([e[1].elapsed_time(e[2]) for e in wl.ev])) if dist is not None else None
    quant_ms = float(np.mean([e[2].elapsed_time(e[3]) for e in wl.ev]))
    rast_ms = float(np.mean([e[3].elapsed_time(e[4]) for e in wl.ev]))

    # stage-only rates: STFT->dB stage and quantise+raster stage, HIP events
    def time_stage(fn, reps=10, spin_ms=40.0):
        t_sp = time.perf_counter()  # (same settling as the headline: the case before this one left the GPU idle for a while)
        while (time.perf_counter() - t_sp) * 1e3 < spin_ms:
            for _ in range(4):
                fn()
            torch.cuda.synchronize(dev)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize(dev)
        return e0.elapsed_time(e1) / reps

    wl.step(dist)
    if wl.use_fused:
        img_ms = time_stage(lambda: ctx.spec_to_img_raster_batch(wl.fused, wl.cmap.data_ptr(), wl.n_colors, d_range=wl.range_db.data_ptr()))
        # the two separate kernels on the same data, for the record (they are what a host that wants only SOME tiles runs)
        quant_ms = time_stage(lambda: ctx.spec_to_img_batch_ranged(wl.imgd, wl.range_db.data_ptr(), 258))
        rast_ms = time_stage(lambda: ctx.raster_tiles(wl.rast, wl.cmap.data_ptr(), wl.n_colors))
    else:
        img_ms = time_stage(lambda: (ctx.spec_to_img_batch_ranged(wl.imgd, wl.range_db.data_ptr(), 258),
                                     ctx.raster_tiles(wl.rast, wl.cmap.data_ptr(), wl.n_colors)))

    # Cold launch: a desktop viewer's add_tracks is always the first launch after an idle gap.  The dominant kernel alone,
    # once after each of five 0.25 s idle gaps (the clocks have fallen back by then), HIP events on the launch stream.
    cold_series = []
    if rank == 0 and world == 1:
        for _ in range(5):
            torch.cuda.synchronize(dev)
            time.sleep(0.25)
            wl.plan.time_kernel(True)
            wl.stft_only()
            torch.cuda.synchronize(dev)
            h = wl.plan.kernel_ms_history()
            if len(h):
                cold_series.append(float(h[-1]))

    # The packed-f32 pipeline of the same kernel (th_plan_set_kernel(plan, 9): v_pk_fma_f32 butterflies on register pairs, 417
    # instead of 681 VALU instructions per frame — the lever VERDICT r3 named), inside the same step, right after the
    # default: reported so that the driver's record shows why it is not the default.
    pk_ms = None
    if rank == 0 and world == 1 and ta.ab_variants():   # (A/B builds only: the product library no longer carries selector 9)
        try:
            wl.plan.set_kernel(9)
            for _ in range(args.spin_up_steps // 2):
                wl.step(dist)
            torch.cuda.synchronize(dev)
            wl.plan.time_kernel(True)
            for _ in range(args.steps):
                wl.step(dist)
            torch.cuda.synchronize(dev)
            pk_ms = float(np.mean(wl.plan.kernel_ms_history()[-args.steps:]))
        except Exception:
            pk_ms = None
        finally:
            wl.plan.set_kernel(args.kernel)
            wl.plan.time_kernel(True)

    # The extras below (other configs and framings, latency, end to end ...) describe ONE GPU: they run at N = 1 only.  At
    # N > 1 rank 0 reports the timed step and the tile gather and every rank leaves right behind them (round 2 ran the
    # extras on rank 0 while the other ranks sat in the final barrier, stretching the N = 8 wall time by seconds).
    extras = rank == 0 and world == 1 and not args.no_single_track

    # waveform side of the path: every decimation level of every channel, one pass over the audio
    wave = None
    if extras:
        from thesia_amd import _ffi
        n_lv = 13
        tot = ta.api.pyramid_offset(n, n_lv)
        pyr = torch.empty((wl.n_tracks, tot), dtype=torch.float32, device=dev)
        pdesc = (_ffi.PyramidDesc * wl.n_tracks)(*[_ffi.PyramidDesc(wl.wav[i].data_ptr(), pyr[i].data_ptr(), n, n_lv, 0)
                                                   for i in range(wl.n_tracks)])
        pyr_ms = time_stage(lambda: ctx.waveform_pyramid_dev(pdesc))
        smp = wl.n_tracks * n
        wave = {"workload": f"levels 0..{n_lv - 1} (min, max, mean) of {wl.n_tracks} channels x {n} samples",
                "ms": pyr_ms, "msamples_per_s": smp / 1e6 / (pyr_ms * 1e-3), "bound": "hbm",
                "algorithmic_bytes_per_sample": 4 + 4.0 * tot / n,
                "frac": (smp * 4 + wl.n_tracks * tot * 4) / (pyr_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del pyr
        # what th_tm_add_tracks builds since round 3: levels 2 .. 12 only (th_pyramid_desc.first_level = 2) — level 0 is
        # (x, x, x) per sample, half of the pyramid's bytes, level 1 a quarter; tiles of both are served from the resident samples
        tot1 = tot - ta.api.pyramid_offset(n, 2)
        pyr1 = torch.empty((wl.n_tracks, tot1), dtype=torch.float32, device=dev)
        pdesc1 = (_ffi.PyramidDesc * wl.n_tracks)(*[_ffi.PyramidDesc(wl.wav[i].data_ptr(), pyr1[i].data_ptr(), n, n_lv, 2)
                                                    for i in range(wl.n_tracks)])
        pyr1_ms = time_stage(lambda: ctx.waveform_pyramid_dev(pdesc1))
        wave["levels_2_up"] = {"workload": f"levels 2..{n_lv - 1} (first_level = 2: levels 0 and 1 are served from the samples)", "ms": pyr1_ms,
                               "msamples_per_s": smp / 1e6 / (pyr1_ms * 1e-3), "algorithmic_bytes_per_sample": 4 + 4.0 * tot1 / n,
                               "frac": (smp * 4 + wl.n_tracks * tot1 * 4) / (pyr1_ms * 1e-3) / 1e9 / HBM_PEAK_GBS}
        del pyr1

    # The other BASELINE configs' STFT stages on this GPU AT THEIR BASELINE.md §3 SIZES, same definition as `roofline`
    # (algorithmic bytes of SURVEY 8(d) / HIP-event duration of the dominant kernel inside the config's own whole step):
    # config 3 (64 stereo 48 kHz tracks x 60 s = 128 channels, n_fft 4096 / hop 1024), config 4 (32 tracks 44.1 kHz x 60 s,
    # n_fft 2048 / hop 512, 128 mels: fused epilogue, and the matrix-core path with its fp32-MFMA fraction) and config 1's
    # shape (n_fft 1024 / hop 256: the single 2 113 529-sample track, and the 128-track batch for the roofline);
    # `other_framings`: the framings the UI reaches at its 40 ms default (t_overlap 1 .. 32 -> tracks.ts:207) and the
    # short / long transforms, 128 channels x 30 s each, 20 launches back to back.
    other, roof_cfg = None, []
    if extras:
        other = []
        sec60 = 2.0 * args.seconds if args.seconds == 30.0 else 60.0
        wav60 = synth_on_gpu(torch, dev, list(range(3000, 3000 + 128)), 48000, int(round(sec60 * 48000)))   # 64 stereo tracks: two seeds each
        wav44 = synth_on_gpu(torch, dev, list(range(2000, 2032)), 44100, int(round(sec60 * 44100)))
        wav_c1 = synth_on_gpu(torch, dev, [4000], 48000, 2113529)  # stand-in for samples/sample_48k.wav (audio.rs:506-508: shape [1, 2113529])
        cases = (("cfg3", "cfg3: 64 stereo 48 kHz tracks x 60 s, n_fft 4096 / hop 1024, linear dB", wav60, 48000, (4096, 1024, 4096, ta.LINEAR, 0), 0),
                 ("cfg4", "cfg4: 32 tracks 44.1 kHz x 60 s, n_fft 2048 / hop 512, mel-128 dB (filterbank fused into the FFT kernel)", wav44, 44100, (2048, 512, 2048, ta.MEL, 128), 0),
                 ("cfg4_mfma", "cfg4 on the matrix cores: same, FFT kernel -> amplitude rows -> mel_mfma_kernel (v_mfma_f32_16x16x4_f32)", wav44, 44100, (2048, 512, 2048, ta.MEL, 128), 3),
                 ("cfg1_one_track", "cfg1 shape, one track: 48 kHz mono 2113529 samples, n_fft 1024 / hop 256, linear dB", wav_c1, 48000, (1024, 256, 1024, ta.LINEAR, 0), 0),
                 ("cfg1_batch", "cfg1 shape, batch: n_fft 1024 / hop 256, linear dB", wl.wav, 48000, (1024, 256, 1024, ta.LINEAR, 0), 0),
                 ("app_default_linear", "app default framing (40 ms, t_overlap 4): 48 kHz, 1920 / 480 / 2048, linear dB", wl.wav, 48000, (1920, 480, 2048, ta.LINEAR, 0), 0),
                 ("app_default_mel", "app default, mel scale (the app's own default: 347 mels)", wl.wav, 48000, (1920, 480, 2048, ta.MEL, 0), 0),
                 ("app_default_mel_one_frame_epilogue", "app default, mel scale, the fused epilogue one frame at a time (selector 13: round 4's form; the default takes frame pairs)", wl.wav, 48000, (1920, 480, 2048, ta.MEL, 0), 13),
                 ("overlap2", "40 ms, t_overlap 2: 1920 / 960 / 2048, linear dB", wl.wav, 48000, (1920, 960, 2048, ta.LINEAR, 0), 0),
                 ("overlap8", "40 ms, t_overlap 8: 1920 / 240 / 2048, linear dB", wl.wav, 48000, (1920, 240, 2048, ta.LINEAR, 0), 0),
                 ("overlap16", "40 ms, t_overlap 16: 1920 / 120 / 2048, linear dB", wl.wav, 48000, (1920, 120, 2048, ta.LINEAR, 0), 0),
                 ("sr44k_default", "44.1 kHz default: 1764 / 441 / 2048, linear dB", wav44, 44100, (1764, 441, 2048, ta.LINEAR, 0), 0),
                 ("sr96k_default", "96 kHz default shape: 3840 / 960 / 4096, linear dB", wl.wav, 96000, (3840, 960, 4096, ta.LINEAR, 0), 0),
                 ("sr96k_default_mel", "96 kHz default, mel scale (404 mels; round 6: the moment form of the filterbank in the FFT kernel's epilogue, one kernel)", wl.wav, 96000, (3840, 960, 4096, ta.MEL, 0), 0),
                 ("sr96k_default_mel_two_kernels", "96 kHz default, mel scale, round 5's route (selector 12: FFT kernel -> amplitude rows -> banded sums, lane = mel)", wl.wav, 96000, (3840, 960, 4096, ta.MEL, 0), 12),
                 ("sr88k_default_mel", "88.2 kHz default, mel scale (432 mels, 3528 / 882 / 4096; moment-form epilogue)", wav44, 88200, (3528, 882, 4096, ta.MEL, 0), 0),
                 ("nfft512", "short transform: n_fft 512 / hop 128, linear dB (four frames per wave)", wl.wav, 48000, (512, 128, 512, ta.LINEAR, 0), 0),
                 ("sr8k_default", "8 kHz default shape: 320 / 80 / 512, linear dB", wl.wav, 8000, (320, 80, 512, ta.LINEAR, 0), 0),
                 ("sr8k_default_mel", "8 kHz default, mel scale (257 mels: banded sums, lane = mel, in the epilogue of the four-frames-per-wave kernel)", wl.wav, 8000, (320, 80, 512, ta.MEL, 0), 0),
                 ("nfft8192", "long transform: n_fft 8192 / hop 2048, linear dB (one workgroup per frame)", wl.wav, 48000, (8192, 2048, 8192, ta.LINEAR, 0), 0),
                 ("nfft16384", "long transform: n_fft 16384 / hop 4096, linear dB (one workgroup per frame)", wl.wav, 48000, (16384, 4096, 16384, ta.LINEAR, 0), 0),
                 ("nfft32768", "very long transform: n_fft 32768 / hop 8192, linear dB (400 ms window at 48 kHz; round 5: sixteen 1024-point wave transforms + one combining pass, stft_subwave_kernel)", wl.wav, 48000, (32768, 8192, 32768, ta.LINEAR, 0), 0),
                 *((("nfft32768_block_kernel", "n_fft 32768 / hop 8192 on the workgroup-per-frame Stockham kernel (selector 14: rounds 3-4's plan; A/B builds only)", wl.wav, 48000, (32768, 8192, 32768, ta.LINEAR, 0), 14),) if ta.ab_variants() else ()),
                 ("nfft65536", "n_fft 65536 / hop 16384, linear dB (1.4 s window at 48 kHz; round 5: the workgroup-per-frame plan with planar LDS exchanges)", wl.wav, 48000, (65536, 16384, 65536, ta.LINEAR, 0), 0),
                 # the Mel default of long windows has more than 512 mels (src-common/src/lib.rs:91-103): two kernels since round 4
                 ("nfft4096_mel_default", "48 kHz, n_fft 4096 / hop 1024, mel scale at the default count (695 mels; round 6: moment-form epilogue, one kernel)", wl.wav, 48000, (4096, 1024, 4096, ta.MEL, 0), 0),
                 ("nfft4096_mel_default_two_kernels", "48 kHz, n_fft 4096 / hop 1024, 695 mels, round 5's route (selector 12: FFT kernel -> amplitude rows -> mel_mfma_kernel)", wl.wav, 48000, (4096, 1024, 4096, ta.MEL, 0), 12),
                 ("nfft8192_mel_default", "48 kHz, n_fft 8192 / hop 2048, mel scale at the default count (1392 mels; round 6: moment-form epilogue in the workgroup-per-frame kernel)", wl.wav, 48000, (8192, 2048, 8192, ta.MEL, 0), 0),
                 ("nfft8192_mel_default_two_kernels", "48 kHz, n_fft 8192 / hop 2048, 1392 mels, round 5's route (selector 12: block kernel -> amplitude rows -> mel_mfma_kernel)", wl.wav, 48000, (8192, 2048, 8192, ta.MEL, 0), 12),
                 ("nfft16384_mel_default", "48 kHz, n_fft 16384 / hop 4096, mel scale at the default count (2785 mels; round 6: moment-form epilogue, one kernel)", wl.wav, 48000, (16384, 4096, 16384, ta.MEL, 0), 0),
                 ("nfft16384_mel_default_two_kernels", "48 kHz, n_fft 16384 / hop 4096, 2785 mels, round 5's route (selector 12)", wl.wav, 48000, (16384, 4096, 16384, ta.MEL, 0), 12),
                 # the UI's own way to n_fft 16384: a 340 ms window is 16320 samples, hop 4080 — not n_fft / 4.  Linear rows run the subwave plan there; a mel plan
                 # takes the workgroup-per-frame kernel for its epilogue (round 6), the two-kernel route stays on the subwave plan
                 ("win340ms_mel_default", "48 kHz, 340 ms window: 16320 / 4080 / 16384, Mel default (2785 mels): block kernel with the moment-form epilogue", wl.wav, 48000, (16320, 4080, 16384, ta.MEL, 0), 0),
                 ("win340ms_mel_default_two_kernels", "48 kHz, 340 ms window, 2785 mels, round 5's route (selector 12: stft_subwave_kernel -> amplitude rows -> mel_mfma_kernel)", wl.wav, 48000, (16320, 4080, 16384, ta.MEL, 0), 12),
                 ("win340ms", "48 kHz, 340 ms window: 16320 / 4080 / 16384, linear dB (stft_subwave_kernel)", wl.wav, 48000, (16320, 4080, 16384, ta.LINEAR, 0), 0),
                 # two more Mel settings the UI's controls produce that ran two kernels until the round's last day (scripts/ui_shapes.py lists all 836):
                 # the 40 ms default at t_overlap 8 / f_overlap 2 (a hop whose grid-aligned frame loop has no mel epilogue: the plain loop + epilogue now), and
                 # 16 kHz audio under an 85 ms window (771 mels at n_fft 2048: more than an LDS table holds — the moment form with its table in global memory)
                 ("tov8_fov2_mel", "48 kHz, 40 ms window, t_overlap 8, f_overlap 2: 1920 / 240 / 4096, Mel default (695 mels), one kernel", wl.wav, 48000, (1920, 240, 4096, ta.MEL, 0), 0),
                 ("tov8_fov2_mel_two_kernels", "the same, selector 12 (grid-aligned frame loop -> amplitude rows -> mel_mfma_kernel)", wl.wav, 48000, (1920, 240, 4096, ta.MEL, 0), 12),
                 ("sr16k_win85ms_mel", "16 kHz, 85 ms window: 1360 / 340 / 2048, Mel default (771 mels), one kernel", wl.wav, 16000, (1360, 340, 2048, ta.MEL, 0), 0),
                 ("sr16k_win85ms_mel_two_kernels", "the same, selector 12 (FFT kernel -> amplitude rows -> mel_mfma_kernel)", wl.wav, 16000, (1360, 340, 2048, ta.MEL, 0), 12))
        for key, label, wav_, sr_, (w_, h_, nf_, scale, n_mel), sel in cases:
            try:
                pl = ta.Plan(ctx, sr_, w_, h_, nf_, scale, n_mel)
                if sel:
                    pl.set_kernel(sel)
                n_ = wav_.shape[1]
                T_, H_ = pl.n_frames(n_), pl.height
                sp_ = ta.pitch_f32(H_)
                spec_ = torch.empty((wav_.shape[0], T_, sp_), dtype=torch.float32, device=dev)
                mm_ = torch.empty((wav_.shape[0], 2), dtype=torch.float32, device=dev)
                ch_ = (ta.ChanDesc * wav_.shape[0])(*[ta.ChanDesc(wav_[i].data_ptr(), spec_[i].data_ptr(), n_, T_, sp_)
                                                      for i in range(wav_.shape[0])])
                pl.time_kernel(True)
                ms_ = time_stage(lambda: pl.calc_spec_batch_dev(ch_, mm_.data_ptr()), 20)
                k_ms = float(np.mean(pl.kernel_ms_history()[-20:]))   # the dominant kernel launch alone
                frames_ = wav_.shape[0] * T_
                bpf = 4 * h_ + 4 * H_
                e = {"key": key, "workload": f"{label}; {wav_.shape[0]} channels x {n_} samples", "kernel": pl.kernel_name, "frames": frames_,
                     "stage_ms": ms_, "avg_launch_ms": k_ms, "frames_per_s": frames_ / (ms_ * 1e-3), "bound": "hbm",
                     "algorithmic_bytes_per_frame": bpf, "achieved": frames_ * bpf / (k_ms * 1e-3) / 1e9, "unit": "GB/s"}
                e["frac"] = e["achieved"] / HBM_PEAK_GBS
                if sel == 3:
                    # the whole two-kernel stage is what this path costs (the event pair only brackets the FFT kernel); the
                    # contraction itself: algorithmic flops = 2 per non-zero filterbank weight per frame (the triangles hold
                    # <= 2 non-zeros per frequency bin), against the dense fp32 MFMA peak
                    fb = ta.calc_mel_fb(sr_, nf_, H_)
                    nnz = int(np.count_nonzero(fb))
                    mel_ms = max(ms_ - k_ms, 1e-6)
                    e.update({"bound": "mfma", "mel_mfma_kernel_ms": mel_ms, "filterbank_nonzeros": nnz,
                              "algorithmic_flops_per_frame": 2 * nnz, "dense_flops_per_frame": 2 * fb.size,
                              "mfma_frac": frames_ * 2 * nnz / (mel_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                              "mfma_frac_dense_equivalent": frames_ * 2 * fb.size / (mel_ms * 1e-3) / 1e12 / MFMA_F32_PEAK_TFLOPS,
                              "peak_TFLOPs": MFMA_F32_PEAK_TFLOPS,
                              "achieved": frames_ * bpf / (ms_ * 1e-3) / 1e9, "avg_launch_ms": ms_})
                    e["frac"] = e["achieved"] / HBM_PEAK_GBS
                elif "+mel_" in pl.kernel_name:  # two kernels: the stage is what the path costs (the event pair brackets the first)
                    e.update({"fft_kernel_ms": k_ms, "avg_launch_ms": ms_, "achieved": frames_ * bpf / (ms_ * 1e-3) / 1e9})
                    e["frac"] = e["achieved"] / HBM_PEAK_GBS
                elif label.startswith("cfg") and wav_.shape[0] > 1:
                    # the BASELINE configs also the way the headline is measured: the kernel's average launch inside this
                    # config's own whole step (STFT -> range -> u16 image -> level-0 RGBA), not 20 launches back to back
                    # (under the power cap what runs next to a kernel decides its clock: `back_to_back_*` keeps the other)
                    wcfg = Workload(torch, ta, ctx, dev, list(range(wav_.shape[0])), sr_, n_, w_, h_, nf_, 0, cmap_bytes,
                                    base=type("B", (), {"wav": wav_, "n_tracks": wav_.shape[0]})(), scale=scale, n_mel=n_mel)
                    t_sp = time.perf_counter()
                    while (time.perf_counter() - t_sp) * 1e3 < 40.0:
                        for _ in range(3):
                            wcfg.step(None)
                        torch.cuda.synchronize(dev)
                    wcfg.plan.time_kernel(True)
                    t_ = time.perf_counter()
                    for _ in range(20):
                        wcfg.step(None, record=True)
                    torch.cuda.synchronize(dev)
                    step_ms = (time.perf_counter() - t_) / 20 * 1e3
                    k_in = float(np.mean(wcfg.plan.kernel_ms_history()[-20:]))
                    e.update({"back_to_back_avg_launch_ms": k_ms, "back_to_back_frac": e["frac"], "avg_launch_ms": k_in,
                              "achieved": frames_ * bpf / (k_in * 1e-3) / 1e9, "step_ms": step_ms,
                              "step_frames_per_s": frames_ / (step_ms * 1e-3), "launch_ms": dist_of(wcfg.plan.kernel_ms_history()[-20:])})
                    e["frac"] = e["achieved"] / HBM_PEAK_GBS
                    del wcfg
                (roof_cfg if label.startswith("cfg") else other).append(e)
                pl.close()
                del spec_
            except Exception as e:  # extras must not break the bench line
                (roof_cfg if label.startswith("cfg") else other).append({"key": key, "workload": label, "error": str(e)[:200]})
        del wav44, wav60, wav_c1

    single = None
    if extras:
        w1 = Workload(torch, ta, ctx, dev, [0], sr, 60 * sr, win, hop, n_fft, args.kernel, cmap_bytes)
        for _ in range(3):
            w1.step(None)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        reps = 50
        for _ in range(reps):
            w1.step(None)
        torch.cuda.synchronize(dev)
        d1 = (time.perf_counter() - t1) / reps
        k1 = time_stage(w1.stft_only, 50)
        single = {"workload": "cfg2: 1 track 48 kHz mono 60 s, n_fft=2048 hop=512, dB + colormap raster",
                  "frames": w1.frames, "ms_per_step": d1 * 1e3, "frames_per_s": w1.frames / d1,
                  "stft_kernel_us": k1 * 1e3, "stft_frames_per_s": w1.frames / (k1 * 1e-3)}
        # the same step replayed from a HIP graph (th_ctx_capture_*): eight small launches, launch-bound
        try:
            w1.plan.time_kernel(False)
            graph = ctx.capture(lambda: w1.step(None))
            for _ in range(3):
                graph.launch()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            for _ in range(reps):
                graph.launch()
            torch.cuda.synchronize(dev)
            dg = (time.perf_counter() - t1) / reps
            single["ms_per_step_hip_graph"] = dg * 1e3
            single["frames_per_s_hip_graph"] = w1.frames / dg
            graph.close()
        except Exception as e:  # report, do not fail the bench line
            single["hip_graph_error"] = str(e)[:200]
        del w1

    # ---- the image-tile gather to the root (outside the timed step, reported separately): every rank's RGBA
    # tile buffer, device to device, one point-to-point transfer per peer posted at once
    gather = None
    if dist is not None:
        from thesia_amd import dist as tdist
        # (one untimed gather first: the first point-to-point transfer between two ranks sets up their RCCL connection)
        # (rehearsal on one card: a 64 MiB slice per rank through host memory — gloo moves CPU tensors)
        payload = wl.rgba.view(-1)[: 1 << 26].cpu() if SHARE_GPU else wl.rgba.view(-1)
        got = tdist.gather_tensor_to_root(payload[: 1 << 20], dist, root=0)
        del got
        barrier()
        t0 = time.perf_counter()
        got = tdist.gather_tensor_to_root(payload, dist, root=0)
        barrier()
        g_dt = time.perf_counter() - t0
        if rank == 0:
            inbound = sum(int(t.numel()) for r, t in enumerate(got) if r != 0)
            gather = {"what": "level-0 RGBA tiles of every track, device-resident send/recv to rank 0 (batch_isend_irecv)",
                      "ms": g_dt * 1e3, "bytes_per_rank": int(payload.numel()), "inbound_bytes_root": inbound,
                      "inbound_GBs": inbound / g_dt / 1e9 if inbound else None, "ranks": world}
        del got

    # ---- PCIe-inclusive end to end + the tile-request latency path, through the TrackManager (th_tm_*): host buffers in,
    # tiles out.  Never `value` (inputs start in host memory here).
    e2e = latency = None
    if extras:
        try:
            e2e, latency = tm_end_to_end_and_latency(torch, ta, ctx, wl, sr, cmap_bytes)
        except Exception as e:  # extras must not break the bench line
            e2e = {"error": str(e)[:300]}

    # ---- strong-scaling anchor: ALL of BASELINE config 5 (1024 tracks x 30 s) resident on this one GPU, same step
    full5 = None
    if extras and not args.no_full_cfg5 and args.tracks_per_gpu == 128:
        try:
            t_s = time.perf_counter()
            del wl.rgba  # (rebuilt below at the larger size; the 128-track line above is complete)
            # (1024 DISTINCT tracks, seeds 0x7E51A + 0 .. 1023 as BASELINE.md prescribes: all 5.9 GB of input are read)
            w5 = Workload(torch, ta, ctx, dev, list(range(1024)), sr, n, win, hop, n_fft, args.kernel, cmap_bytes)
            for _ in range(2):
                w5.step(None)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            reps5 = 5
            for _ in range(reps5):
                w5.step(None)
            torch.cuda.synchronize(dev)
            d5 = (time.perf_counter() - t1) / reps5
            full5 = {"workload": "cfg5 complete on ONE GPU: 1024 distinct tracks x 30 s 48 kHz mono, n_fft=2048 hop=512, same step",
                     "frames": w5.frames, "ms_per_step": d5 * 1e3, "frames_per_s": w5.frames / d5,
                     "resident_GB": (w5.spec.numel() * 4 + w5.img.numel() * 2 + w5.rgba.numel() + w5.wav.numel() * 4) / 1e9,
                     "setup_s": time.perf_counter() - t_s}
            del w5
        except Exception as e:
            full5 = {"error": str(e)[:300]}

    if rank == 0:
        total_frames = wl.frames * world
        bytes_per_frame = 4 * hop + 4 * wl.H            # read 4*hop + write 4*H
        # frames one launch of the dominant kernel processes: all of them (the wave kernel also takes the 4 boundary
        # frames per channel, as one-frame chunks with a reflect-indexed fetch)
        interior = wl.frames
        ach = interior * bytes_per_frame / (stft_ms * 1e-3) / 1e9
        # PMC-measured HBM bytes per launch cannot be collected inside this run (separate rocprofv3 --pmc passes): a static
        # figure with its provenance, not a run value
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "stft_hbm_traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                traffic = tj.get("bytes_per_launch")
                traffic_source = f"profiles/stft_hbm_traffic.json: PMC FETCH_SIZE x2 + WRITE_SIZE, separate passes, {tj.get('date', '?')}, commit {tj.get('commit', '?')}"
            except Exception:
                traffic = None
        # a measured device-to-device copy on this box in the same run, beside the vendor peak — the
        # library's own 16-byte-per-lane streaming kernel (th_dev_copy), 1 GiB each way
        copy_gbs = None
        try:
            a_ = torch.empty(1 << 28, dtype=torch.float32, device=dev)  # 1 GiB
            b_ = torch.empty_like(a_)
            nb_ = a_.numel() * 4
            for _ in range(2):
                ctx.dev_copy(b_.data_ptr(), a_.data_ptr(), nb_)
            c0, c1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            c0.record()
            for _ in range(5):
                ctx.dev_copy(b_.data_ptr(), a_.data_ptr(), nb_)
            c1.record()
            torch.cuda.synchronize()
            copy_gbs = 5 * 2 * nb_ / (c0.elapsed_time(c1) * 1e-3) / 1e9
            del a_, b_
        except Exception:
            copy_gbs = None
        # the kernel's memory skeleton on this box, in this run (N = 1 only: a child process with its own buffers)
        skeleton = None
        if world == 1 and not args.no_skeleton:
            skeleton = memory_skeleton()
            try:
                sk = skeleton["gaps_1ms"]["no_work_ms"]
                skeleton["kernel_over_skeleton"] = stft_ms / sk
                skeleton["kernel_median_over_skeleton"] = float(np.median(stft_series)) / sk
                skeleton["frac_if_kernel_ran_at_skeleton"] = interior * bytes_per_frame / (sk * 1e-3) / 1e9 / HBM_PEAK_GBS
            except Exception:
                pass
        out = {
            "metric": "STFT frames/sec (whole step: STFT->dB->min/max->u16 image->level-0 RGBA raster), n_fft=2048",
            "value": total_frames * args.steps / dt, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "spin_up_steps": args.spin_up_steps, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"cfg5 shard: {args.tracks_per_gpu} tracks/GPU x {args.seconds:g} s 48 kHz mono, "
                                   f"n_fft={n_fft} hop={hop} Hann, linear dB + u16 image + level-0 RGBA tiles",
                       "tracks_per_gpu": args.tracks_per_gpu, "frames_per_gpu": wl.frames,
                       "parallelism": f"track-sharded x{world}, 2-float dB-range all-reduce"
                                      + (" — REHEARSAL: all ranks share GPU 0, gloo (TH_BENCH_SHARE_GPU=1); not a scaling figure" if SHARE_GPU else ""),
                       **({"ranks_share_one_gpu": True, "backend": "gloo"} if SHARE_GPU else {})},
            "stft_frames_per_s": total_frames / (stft_stage_ms * 1e-3),
            "raster_mpixels_per_s": wl.pixels * world / 1e6 / (img_ms * 1e-3),
            "stft_kernel": wl.plan.kernel_name,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": ach / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "kernel": wl.plan.kernel_name, "avg_launch_ms": stft_ms, "launch_ms": dist_of(stft_series),
                         "launch_ms_series": [round(v, 4) for v in stft_series],
                         "frac_at_median_launch": interior * bytes_per_frame / (float(np.median(stft_series)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "frac_at_min_launch": interior * bytes_per_frame / (min(stft_series) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "memory_skeleton": skeleton,
                         "stage_ms_incl_init_and_boundary_frames": stft_stage_ms,
                         "algorithmic_bytes_per_frame": bytes_per_frame, "frames_per_launch": interior,
                         "read_only_frac": interior * 4 * hop / (stft_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                         "measured_copy_GBs": copy_gbs, "measured_copy_kernel": "th_dev_copy (16 B per lane, 1 GiB -> 1 GiB)",
                         "frac_of_measured_copy": (ach / copy_gbs) if copy_gbs else None},
            # the two other kernels of the step, same definition (algorithmic bytes / HIP-event duration)
            "roofline_other": [
                {"kernel": "spec_to_img_kernel", "bound": "hbm", "avg_launch_ms": quant_ms,
                 "algorithmic_bytes_per_pixel": 6, "achieved": wl.pixels * 6 / (quant_ms * 1e-3) / 1e9,
                 "frac": wl.pixels * 6 / (quant_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"},
                {"kernel": "raster_level0_kernel", "bound": "hbm", "avg_launch_ms": rast_ms,
                 "algorithmic_bytes_per_pixel": 6, "achieved": wl.n_tracks * wl.tile_px * 6 / (rast_ms * 1e-3) / 1e9,
                 "frac": wl.n_tracks * wl.tile_px * 6 / (rast_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "unit": "GB/s"}] + roof_cfg,
        }
        # ---- what must survive the driver's reader goes into `roofline` as FLAT SCALARS (the driver keeps the scalar keys of
        # `roofline` and an 8 KB tail of stdout); the nested detail goes out on an EARLIER line and into a side file
        rf = out["roofline"]
        ld = rf.pop("launch_ms")
        series = rf.pop("launch_ms_series")
        rf.pop("memory_skeleton")
        rf.update({"launch_ms_min": ld["min"], "launch_ms_median": ld["median"], "launch_ms_p90": ld["p90"], "launch_ms_max": ld["max"],
                   "first_timed_launch_ms": series[0] if series else None})
        if cold_series:
            cm = float(np.median(cold_series))
            rf.update({"cold_first_launch_ms": cm, "cold_first_launch_max_ms": max(cold_series),
                       "cold_first_launch_frac": interior * bytes_per_frame / (cm * 1e-3) / 1e9 / HBM_PEAK_GBS})
        if pk_ms is not None:
            rf["packed_f32_variant_ms"] = pk_ms
            rf["packed_f32_variant_frac"] = interior * bytes_per_frame / (pk_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
        if skeleton is not None:
            for k_src, k_dst in (("gaps_1ms", "skeleton_no_work_ms"), ("back_to_back", "skeleton_no_work_back_to_back_ms")):
                try:
                    rf[k_dst] = skeleton[k_src]["no_work_ms"]
                except Exception:
                    pass
            try:
                rf["skeleton_with_kernel_amount_of_work_ms"] = skeleton["gaps_1ms"]["with_kernel_amount_of_work_ms"]
            except Exception:
                pass
            for k in ("kernel_over_skeleton", "kernel_median_over_skeleton", "frac_if_kernel_ran_at_skeleton"):
                if k in skeleton:
                    rf[k] = skeleton[k]
        rf.update({"image_stage_in_step_ms": image_ms, "image_stage_fused": bool(wl.use_fused),
                   "image_stage_bytes_per_pixel": 10 if wl.use_fused else 12,
                   "image_stage_frac": wl.pixels * (10 if wl.use_fused else 12) / (image_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "quantise_ms": quant_ms, "quantise_frac": out["roofline_other"][0]["frac"],
                   "raster_ms": rast_ms, "raster_frac": out["roofline_other"][1]["frac"],
                   "stft_frames_per_s": out["stft_frames_per_s"], "raster_mpixels_per_s": out["raster_mpixels_per_s"]})
        for e in roof_cfg + (other or []):
            k = e.get("key")
            if not k or "error" in e:
                continue
            rf[k + "_ms"] = e.get("avg_launch_ms")
            rf[k + "_frac"] = e.get("frac")
            if "back_to_back_frac" in e:
                rf[k + "_back_to_back_frac"] = e["back_to_back_frac"]
            if "step_ms" in e:
                rf[k + "_step_ms"] = e["step_ms"]
            if "mfma_frac" in e:
                rf[k + "_mfma_frac"] = e["mfma_frac"]
                rf[k + "_contraction_ms"] = e["mel_mfma_kernel_ms"]
        if full5 is not None and "ms_per_step" in full5:
            rf["cfg5_full_1gpu_ms_per_step"] = full5["ms_per_step"]
            rf["cfg5_full_1gpu_frames_per_s"] = full5["frames_per_s"]
        if single is not None:
            rf["cfg2_single_track_ms_per_step"] = single["ms_per_step"]
            rf["cfg2_single_track_stft_kernel_us"] = single["stft_kernel_us"]
        if wave is not None:
            rf["waveform_pyramid_ms"] = wave["ms"]
            rf["waveform_pyramid_frac"] = wave["frac"]
            rf["waveform_pyramid_levels_2_up_ms"] = wave["levels_2_up"]["ms"]
        if e2e is not None and "upload_pyramid_stft_ms" in e2e:
            rf["e2e_pcie_frames_per_s_compute_only"] = e2e["frames_per_s_compute_only"]
            rf["e2e_pcie_frames_per_s_with_batched_tile_fetch"] = e2e.get("frames_per_s_with_batched_tile_fetch")
            rf["set_dB_range_32_tracks_ms"] = e2e["set_dB_range_ms"]
        if latency is not None:
            for k in ("spec_level0", "spec_lod_1_0", "waveform"):
                try:
                    rf["tile_" + k + "_p50_us"] = latency[k]["threads_1"]["p50_us"]
                except Exception:
                    pass
        if gather is not None:
            rf["tile_gather_ms"] = gather["ms"]
            rf["tile_gather_inbound_GBs"] = gather["inbound_GBs"]
            rf["tile_gather_ranks"] = gather["ranks"]
        if range_exchange_ms is not None:
            rf["range_allreduce_in_step_ms"] = range_exchange_ms
            rf["range_allreduce_ranks"] = world
        for k, v in list(rf.items()):  # 5 significant digits are plenty and keep the line short
            if isinstance(v, float):
                rf[k] = float(f"{v:.5g}")
        extras_out = {"extras_of": "bench.py (nested detail of the last line's flat `roofline` scalars)",
                      "launch_ms_series": series, "cold_launch_ms_series": cold_series, "memory_skeleton": skeleton,
                      "roofline_other": out.pop("roofline_other")}
        for k, v in (("tile_gather", gather), ("end_to_end_pcie_inclusive", e2e), ("tile_latency", latency),
                     ("strong_scaling_anchor_cfg5_1gpu", full5), ("single_track_cfg2", single), ("waveform_pyramid", wave),
                     ("other_framings", other)):
            if v is not None:
                extras_out[k] = v
        if gather is not None:
            out["tile_gather"] = {"ms": gather["ms"], "inbound_GBs": gather["inbound_GBs"], "ranks": gather["ranks"]}
        if range_agrees is not None:
            out["global_dB_range"] = {"min_max_dB": global_range, "identical_on_all_ranks": range_agrees, "ranks": world}
            if not range_agrees:
                raise SystemExit("bench.py: the ranks disagree on the global dB range — the range exchange is broken")
        if range_exchange_ms is not None:
            out["range_allreduce"] = {"in_step_ms": range_exchange_ms, "ranks": world, "backend": "gloo via host memory (rehearsal)" if SHARE_GPU else "nccl (RCCL)",
                                      "what": "minmax fold kernel + 2-float all_reduce(MIN) + clamp kernel, HIP events inside the timed step"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cmap_bytes, sr, n, win, hop, n_fft)
        try:  # RCCL prints a version banner through C stdio; push it out first so that the JSON is the last stdout line
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        try:  # side file (scratch on the GPU box; collect_profiles.sh copies it into profiles/)
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "bench_extras.json"), "w") as fh:
                json.dump(extras_out, fh)
        except Exception:
            pass
        print(json.dumps({"bench_extras": extras_out}), flush=True)   # earlier line: the nested detail
        print(json.dumps(out), flush=True)                            # LAST line: the compact record
    barrier()
    del wl
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
