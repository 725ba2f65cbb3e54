"""Thin Python face of the C ABI (include/thesia_amd.h) used by tests and bench.py.

The host logic lives in C++ behind the C ABI (thesia_amd/csrc); this file only marshals numpy
arrays / device pointers.  Names follow the reference: SpecSetting, SpectrogramAnalyzer plan,
TrackManager, encode_*_tile (src-tauri/src/core/{spectrogram,mod,render_tiles}.rs).
"""
from __future__ import annotations

import ctypes as C
from typing import Iterable, Optional, Sequence

import struct
import threading

import numpy as np

from . import _ffi
from ._ffi import (ChanDesc, ImgDesc, RasterDesc, ThError, TileGeom, WaveDesc, c_f32p, c_szp, c_u8p, c_u16p, check,
                   lib, vp)

LINEAR, MEL = 0, 1
WAVEFORM_TILE_MAX_BYTES = 24 + 1024 * 12
SPECTROGRAM_TILE_MAX_BYTES = 40 + 520 * 520 * 4


def _f32(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a: np.ndarray, t):
    return a.ctypes.data_as(t)


# ------------------------------------------------------------------ host-only helpers
def device_count() -> int:
    n = C.c_int(0)
    check(lib.th_device_count(C.byref(n)))
    return n.value


def calc_framing_params(win_ms: float, t_overlap: int, f_overlap: int, sr: int):
    """SpecSetting::calc_framing_params (spectrogram.rs:56-98) -> (hop, win, n_fft)."""
    h, w, n = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(lib.th_calc_framing_params(win_ms, t_overlap, f_overlap, sr, C.byref(h), C.byref(w), C.byref(n)))
    return h.value, w.value, n.value


def stft_n_frames(n_samples: int, win: int, hop: int) -> int:
    t = C.c_size_t()
    check(lib.th_stft_n_frames(n_samples, win, hop, C.byref(t)))
    return t.value


def calc_normalized_win(win: int, n_fft: int) -> np.ndarray:
    out = np.empty(win, np.float32)
    check(lib.th_calc_normalized_win(win, n_fft, _ptr(out, c_f32p)))
    return out


def calc_mel_fb(sr: int, n_fft: int, n_mel: int, fmin: float = 0.0, fmax: Optional[float] = None,
                do_norm: bool = True) -> np.ndarray:
    out = np.empty((n_fft // 2 + 1, n_mel), np.float32)
    check(lib.th_calc_mel_fb(sr, n_fft, n_mel, fmin, -1.0 if fmax is None else fmax, int(do_norm),
                             _ptr(out, c_f32p)))
    return out


def mel_default_n_mel(sr: int, n_fft: int) -> int:
    n = C.c_size_t()
    check(lib.th_mel_default_n_mel(sr, n_fft, C.byref(n)))
    return n.value


def hz_range_to_idx(freq_scale: int, hz_range, sr: int, n: int):
    a, b = C.c_size_t(), C.c_size_t()
    check(lib.th_hz_range_to_idx(freq_scale, hz_range[0], hz_range[1], sr, n, C.byref(a), C.byref(b)))
    return a.value, b.value


def global_db_range(mins, maxs, dB_range: float = 100.0):
    mins, maxs = _f32(mins).ravel(), _f32(maxs).ravel()
    lo, hi = C.c_float(), C.c_float()
    check(lib.th_global_db_range(_ptr(mins, c_f32p), _ptr(maxs, c_f32p), mins.size, dB_range, C.byref(lo),
                                 C.byref(hi)))
    return lo.value, hi.value


def pitch_f32(row_elems: int) -> int:
    """Recommended floats per row for device-resident specs (rows padded to 128 B)."""
    return lib.th_pitch_f32(row_elems)


def pitch_u16(row_elems: int) -> int:
    """Recommended u16 per row for device-resident images (rows padded to 128 B)."""
    return lib.th_pitch_u16(row_elems)


def pyramid_bins(n_samples: int, level: int) -> int:
    return lib.th_waveform_pyramid_bins(n_samples, level)


def pyramid_offset(n_samples: int, level: int) -> int:
    """float offset of level `level` in a channel's pyramid buffer (= total floats for level = n_levels)"""
    return lib.th_waveform_pyramid_offset(n_samples, level)


def ab_variants() -> bool:
    """True when the loaded library was built with -DTH_AB_VARIANTS=1 (the measured-and-dropped kernel variants behind
    th_plan_set_kernel: include/thesia_amd_testing.h); the product build answers False and refuses those selectors."""
    return bool(lib.th_build_ab_variants())


def shard_assign(weights, world: int):
    """th_shard_assign: owner rank of every (track, channel) unit, by frame-count weight."""
    w = np.ascontiguousarray(weights, dtype=np.uint64)
    owner = np.empty(w.size, np.uint32)
    check(lib.th_shard_assign(w.ctypes.data_as(C.POINTER(C.c_uint64)), w.size, world,
                              owner.ctypes.data_as(C.POINTER(C.c_uint32))))
    return owner


def spectrogram_tile_geometry(img_width: int, img_height: int, level_x: int, level_y: int, tile_x: int,
                              tile_y: int) -> TileGeom:
    g = TileGeom()
    check(lib.th_spectrogram_tile_geometry(img_width, img_height, level_x, level_y, tile_x, tile_y, C.byref(g)))
    return g


def waveform_tile_geometry(n_samples: int, level: int, tile_index: int):
    s, b, p = C.c_size_t(), C.c_size_t(), C.c_size_t()
    check(lib.th_waveform_tile_geometry(n_samples, level, tile_index, C.byref(s), C.byref(b), C.byref(p)))
    return s.value, b.value, p.value


# ------------------------------------------------------------------ device context
class DeviceBuffer:
    """hipMalloc'd buffer owned through the C ABI (th_dev_alloc / th_dev_free)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx, self.nbytes = ctx, int(nbytes)
        p = vp()
        check(lib.th_dev_alloc(ctx.handle, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, a: np.ndarray) -> "DeviceBuffer":
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(lib.th_dev_upload(self.ctx.handle, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype)
        assert out.nbytes <= self.nbytes
        check(lib.th_dev_download(self.ctx.handle, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            check(lib.th_dev_free(self.ctx.handle, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Graph:
    """A captured launch sequence (Context.capture); launch() replays it on the context's stream."""

    def __init__(self, handle):
        self.handle = handle

    def launch(self):
        check(lib.th_graph_launch(self.handle))

    def close(self):
        if self.handle:
            lib.th_graph_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Context:
    """th_ctx: one GPU + one HIP stream.  `stream=None` creates a private stream; an int is used as
    the raw hipStream_t (e.g. torch.cuda.current_stream().cuda_stream; 0 = legacy default stream)."""

    def __init__(self, device: int = 0, stream: Optional[int] = None):
        h = vp()
        if stream is None:
            check(lib.th_ctx_create(device, None, C.byref(h)))
        else:
            check(lib.th_ctx_create_ex(device, vp(stream), 1, C.byref(h)))
        self.handle = h
        self.device = device

    def close(self):
        if self.handle:
            check(lib.th_ctx_destroy(self.handle))
            self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def synchronize(self):
        check(lib.th_ctx_synchronize(self.handle))

    def capture(self, fn) -> "Graph":
        """Record the library calls `fn()` makes on this context's stream into a HIP graph (th_ctx_capture_begin / _end).
        Call `fn()` once normally first: tables and scratch buffers are uploaded / sized on first use."""
        check(lib.th_ctx_capture_begin(self.handle))
        h = C.c_void_p()
        try:
            fn()
        except BaseException:
            lib.th_ctx_capture_end(self.handle, C.byref(h))  # leave capture mode; the error of fn() is the one to report
            if h.value:
                lib.th_graph_destroy(h)
            raise
        check(lib.th_ctx_capture_end(self.handle, C.byref(h)))
        return Graph(h)

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def to_device(self, a: np.ndarray) -> DeviceBuffer:
        a = np.ascontiguousarray(a)
        return DeviceBuffer(self, max(a.nbytes, 1)).upload(a)

    def dev_copy(self, dst_ptr: int, src_ptr: int, nbytes: int):
        """Stream-ordered device copy with the library's 16-byte-per-lane streaming kernel (the bench's bandwidth yardstick)."""
        check(lib.th_dev_copy(self.handle, dst_ptr, src_ptr, nbytes))

    def timer_start(self):
        check(lib.th_timer_start(self.handle))

    def timer_stop_ms(self) -> float:
        ms = C.c_float()
        check(lib.th_timer_stop_ms(self.handle, C.byref(ms)))
        return ms.value

    # ---- convert_spectrogram_to_img (drawing.rs:4-33), numpy in/out
    def spec_to_img(self, spec: np.ndarray, i_freq_range, dB_range, colormap_length: Optional[int]) -> np.ndarray:
        spec = _f32(spec)
        T, H = spec.shape
        i0, i1 = i_freq_range
        d_spec = self.to_device(spec)
        d_img = self.alloc(max((i1 - i0) * T * 2, 1))
        check(lib.th_spec_to_img_dev(self.handle, d_spec.ptr, T, H, i0, i1, dB_range[0], dB_range[1],
                                     0 if colormap_length is None else colormap_length, d_img.ptr))
        out = d_img.download((i1 - i0, T), np.uint16)
        d_spec.free()
        d_img.free()
        return out

    def spec_to_img_batch(self, descs: Sequence[ImgDesc], min_dB: float, max_dB: float, colormap_length: int):
        arr = descs if isinstance(descs, C.Array) else (ImgDesc * len(descs))(*descs)
        check(lib.th_spec_to_img_batch_dev(self.handle, arr, len(arr), min_dB, max_dB, colormap_length))

    # ---- encode_spectrogram_tile (render_tiles.rs:281-352), numpy image in, bytes out
    def encode_spectrogram_tile(self, img: np.ndarray, colormap_rgba: bytes, revision: int, level_x: int,
                                level_y: int, tile_x: int, tile_y: int) -> bytes:
        img = np.ascontiguousarray(img, dtype=np.uint16)
        d_img = self.to_device(img)
        try:
            return self.encode_spectrogram_tile_dev(d_img.ptr, img.shape[0], img.shape[1], colormap_rgba, revision,
                                                    level_x, level_y, tile_x, tile_y, 0)
        finally:
            d_img.free()

    def encode_spectrogram_tile_dev(self, d_img: int, img_height: int, img_width: int, colormap_rgba: bytes,
                                    revision: int, level_x: int, level_y: int, tile_x: int, tile_y: int,
                                    img_pitch: int = 0) -> bytes:
        cm = np.frombuffer(bytes(colormap_rgba), np.uint8)
        out = np.empty(SPECTROGRAM_TILE_MAX_BYTES, np.uint8)
        n = C.c_size_t()
        check(lib.th_encode_spectrogram_tile_dev(self.handle, d_img, img_height, img_width, img_pitch, _ptr(cm, c_u8p), cm.size,
                                                 revision, level_x, level_y, tile_x, tile_y, _ptr(out, c_u8p),
                                                 out.size, C.byref(n)))
        return out[: n.value].tobytes()

    def raster_tiles(self, descs: Sequence[RasterDesc], d_colormap: int, n_colors: int):
        arr = descs if isinstance(descs, C.Array) else (RasterDesc * len(descs))(*descs)
        check(lib.th_raster_tiles_dev(self.handle, arr, len(arr), d_colormap, n_colors))

    def make_img_tiles_descs(self, items):
        """items: iterable of (ImgDesc, [device pointer of tile (tx, ty) at tx * n_ty + ty, 0 = skip]) -> ctypes array of
        th_img_tiles_desc (keeps the pointer arrays alive)."""
        items = list(items)
        arr = (_ffi.ImgTilesDesc * len(items))()
        keep = []
        for i, (d, ptrs) in enumerate(items):
            out_h, w = int(d.i_end - d.i_start), int(d.n_frames)
            n_tx, n_ty = (-(-w // 512), -(-out_h // 512)) if out_h and w else (0, 0)
            assert len(ptrs) == n_tx * n_ty, (len(ptrs), n_tx, n_ty)
            pa = (C.c_void_p * max(len(ptrs), 1))(*[C.c_void_p(p or None) for p in ptrs])
            keep.append(pa)
            arr[i].img = d
            arr[i].tiles = C.cast(pa, C.POINTER(C.c_void_p))
            arr[i].n_tiles_x, arr[i].n_tiles_y = n_tx, n_ty
        arr._keep = keep
        return arr

    def spec_to_img_raster_batch(self, descs, d_colormap: int, n_colors: int, min_dB: float = 0.0, max_dB: float = 0.0,
                                 d_range: int = 0):
        """th_spec_to_img_raster_batch_dev: quantise + every level-0 tile in one pass.  descs: make_img_tiles_descs(...)."""
        check(lib.th_spec_to_img_raster_batch_dev(self.handle, descs, len(descs), min_dB, max_dB, d_range or None, d_colormap, n_colors))

    # ---- encode_waveform_tile (render_tiles.rs:232-279)
    def encode_waveform_tile(self, wav: np.ndarray, revision: int, level: int, tile_index: int) -> bytes:
        wav = _f32(wav)
        d = self.to_device(wav)
        try:
            return self.encode_waveform_tile_dev(d.ptr, wav.size, revision, level, tile_index)
        finally:
            d.free()

    def encode_waveform_tile_dev(self, d_wav: int, n_samples: int, revision: int, level: int,
                                 tile_index: int) -> bytes:
        out = np.empty(WAVEFORM_TILE_MAX_BYTES, np.uint8)
        n = C.c_size_t()
        check(lib.th_encode_waveform_tile_dev(self.handle, d_wav, n_samples, revision, level, tile_index,
                                              _ptr(out, c_u8p), out.size, C.byref(n)))
        return out[: n.value].tobytes()

    def waveform_tiles(self, descs: Sequence[WaveDesc]):
        arr = descs if isinstance(descs, C.Array) else (WaveDesc * len(descs))(*descs)
        check(lib.th_waveform_tiles_dev(self.handle, arr, len(arr)))

    def minmax_reduce_dev(self, d_minmax: int, n_chan: int, d_out: int):
        """[min, -max] over n_chan per-channel (min, max) pairs, device to device (core/mod.rs:169-178)"""
        check(lib.th_minmax_reduce_dev(self.handle, d_minmax, n_chan, d_out))

    def global_db_range_dev(self, d_min_negmax: int, dB_range: float, d_range: int):
        """[min, -max] -> [min_dB, max_dB] on the device (core/mod.rs:179-180)"""
        check(lib.th_global_db_range_dev(self.handle, d_min_negmax, dB_range, d_range))

    def minmax_reduce_range_dev(self, d_minmax: int, n_chan: int, dB_range: float, d_range: int, d_min_negmax: int = 0):
        """th_minmax_reduce_dev + th_global_db_range_dev in one launch (single GPU: no all-reduce in between)."""
        check(lib.th_minmax_reduce_range_dev(self.handle, d_minmax, n_chan, dB_range, d_min_negmax or None, d_range))

    def spec_to_img_batch_ranged(self, descs, d_range: int, colormap_len: int):
        arr = descs if isinstance(descs, C.Array) else (ImgDesc * len(descs))(*descs)
        check(lib.th_spec_to_img_batch_dev_ranged(self.handle, arr, len(arr), d_range, colormap_len))

    # ---- channel statistics (sum_squares / abs_max, simd.rs:113-183)
    def channel_stats_dev(self, descs):
        """-> (sum_squares[n], abs_max[n]) float32 host arrays"""
        arr = descs if isinstance(descs, C.Array) else (_ffi.StatsDesc * len(descs))(*descs)
        ss, pk = np.empty(len(arr), np.float32), np.empty(len(arr), np.float32)
        check(lib.th_channel_stats_dev(self.handle, arr, len(arr), _ptr(ss, c_f32p), _ptr(pk, c_f32p)))
        return ss, pk

    # ---- waveform pyramid: all levels of a channel in one pass over the audio
    def waveform_pyramid_dev(self, descs):
        arr = descs if isinstance(descs, C.Array) else (_ffi.PyramidDesc * len(descs))(*descs)
        check(lib.th_waveform_pyramid_dev(self.handle, arr, len(arr)))

    def waveform_pyramid(self, wav: np.ndarray, n_levels: int):
        """Host convenience (tests): returns [level] -> ndarray [bins, 3] of (min, max, mean)."""
        wav = _f32(wav)
        total = pyramid_offset(wav.size, n_levels)
        d = self.to_device(wav)
        o = self.alloc(max(total, 1) * 4)
        try:
            self.waveform_pyramid_dev([_ffi.PyramidDesc(d.ptr, o.ptr, wav.size, n_levels, 0)])
            flat = o.download((max(total, 1),), np.float32)[:total]
        finally:
            d.free()
            o.free()
        return [flat[pyramid_offset(wav.size, l): pyramid_offset(wav.size, l) + 3 * pyramid_bins(wav.size, l)].reshape(-1, 3)
                for l in range(n_levels)]


# ------------------------------------------------------------------ SpectrogramAnalyzer plan
class Plan:
    """th_plan: device-resident window / twiddles / mel filterbank for one framing key
    (SpectrogramAnalyzer::prepare, spectrogram.rs:116-154)."""

    def __init__(self, ctx: Context, sr: int, win: int, hop: int, n_fft: int, freq_scale: int = LINEAR,
                 n_mel: int = 0):
        h = vp()
        check(lib.th_plan_create(ctx.handle, sr, win, hop, n_fft, freq_scale, n_mel, C.byref(h)))
        self.handle, self.ctx = h, ctx
        self.sr, self.win, self.hop, self.n_fft, self.freq_scale = sr, win, hop, n_fft, freq_scale
        f, hh = C.c_size_t(), C.c_size_t()
        check(lib.th_plan_dims(h, C.byref(f), C.byref(hh)))
        self.n_freq, self.height = f.value, hh.value

    def close(self):
        if self.handle:
            check(lib.th_plan_destroy(self.handle))
            self.handle = None

    def time_kernel(self, enable: bool = True):
        """record HIP events around the dominant kernel of every calc_spec_batch_dev (measurement hook)"""
        check(lib.th_plan_time_kernel(self.handle, int(enable)))

    def kernel_ms_history(self) -> np.ndarray:
        """durations (ms) of the most recent timed launches, oldest first (at most 64)"""
        out, n = np.empty(64, np.float32), C.c_size_t()
        check(lib.th_plan_kernel_ms_history(self.handle, _ptr(out, c_f32p), out.size, C.byref(n)))
        return out[: n.value].copy()

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        check(lib.th_plan_last_kernel_ms(self.handle, C.byref(ms)))
        return ms.value

    def mel_moments_info(self) -> dict:
        """the moment-form table of the fused mel epilogue (n_fft 4096 / 8192 / 16384), if the plan has one (groups == 0: it does not)"""
        g, t, d, a = C.c_uint32(), C.c_uint32(), C.c_double(), C.c_double()
        check(lib.th_plan_mel_moments_info(self.handle, C.byref(g), C.byref(t), C.byref(d), C.byref(a)))
        return {"groups": g.value, "taps": t.value, "max_dev": d.value, "max_amp": a.value}

    def set_kernel(self, which: int):
        check(lib.th_plan_set_kernel(self.handle, which))

    @property
    def kernel_name(self) -> str:
        return (lib.th_plan_kernel_name(self.handle) or b"").decode()

    def n_frames(self, n_samples: int) -> int:
        return stft_n_frames(n_samples, self.win, self.hop)

    def calc_spec_batch_dev(self, descs: Sequence[ChanDesc], d_minmax: Optional[int]):
        """Stream-ordered batched calc_spec on device pointers (no sync)."""
        arr = descs if isinstance(descs, C.Array) else (ChanDesc * len(descs))(*descs)
        check(lib.th_calc_spec_batch_dev(self.handle, arr, len(arr), d_minmax))

    def calc_spec_batch_ranged_dev(self, descs: Sequence[ChanDesc], d_minmax: int, dB_range: float, d_range: int):
        """calc_spec_batch_dev + the global dB range of the batch into d_range (single GPU), one call."""
        arr = descs if isinstance(descs, C.Array) else (ChanDesc * len(descs))(*descs)
        check(lib.th_calc_spec_batch_ranged_dev(self.handle, arr, len(arr), d_minmax, dB_range, d_range))

    def calc_spec(self, wav: np.ndarray):
        """calc_spec for one channel with host buffers -> (T x H f32 dB, min, max)."""
        wav = _f32(wav)
        T = self.n_frames(wav.size)
        out = np.empty((T, self.height), np.float32)
        n, mn, mx = C.c_size_t(), C.c_float(), C.c_float()
        check(lib.th_calc_spec_host(self.handle, _ptr(wav, c_f32p), wav.size, _ptr(out, c_f32p), out.size,
                                    C.byref(n), C.byref(mn), C.byref(mx)))
        assert n.value == T
        return out, mn.value, mx.value

    def calc_spec_batch(self, wavs: Iterable[np.ndarray]):
        """Batched calc_spec over host arrays -> (list of specs, (n,2) min/max)."""
        wavs = [_f32(w) for w in wavs]
        ctx = self.ctx
        d_w = [ctx.to_device(w) for w in wavs]
        Ts = [self.n_frames(w.size) for w in wavs]
        d_s = [ctx.alloc(max(T * self.height * 4, 4)) for T in Ts]
        d_mm = ctx.alloc(8 * len(wavs))
        descs = [ChanDesc(dw.ptr, ds.ptr, w.size, T, 0) for dw, ds, w, T in zip(d_w, d_s, wavs, Ts)]
        self.calc_spec_batch_dev(descs, d_mm.ptr)
        ctx.synchronize()
        specs = [ds.download((T, self.height), np.float32) for ds, T in zip(d_s, Ts)]
        mm = d_mm.download((len(wavs), 2), np.float32)
        for b in d_w + d_s + [d_mm]:
            b.free()
        return specs, mm


# ------------------------------------------------------------------ TrackManager mirror
class TileCache:
    """th_tile_cache: mirror of RenderTileCache (render_tiles.rs:51-230) — byte-budgeted LRU of encoded waveform
    tiles plus the waveform / spectrogram revision counters.  Host only."""

    def __init__(self, budget_bytes: int = 0, _borrowed=None):
        if _borrowed is not None:
            self.handle, self._own = _borrowed, False
            return
        h = vp()
        check(lib.th_tile_cache_create(budget_bytes, C.byref(h)))
        self.handle, self._own = h, True

    def close(self):
        if self.handle and self._own:
            check(lib.th_tile_cache_destroy(self.handle))
        self.handle = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def lookup(self, track_id: int, ch: int, level: int, tile_index: int):
        """-> (current waveform revision, bytes or None); a hit makes the entry most recently used."""
        cap = WAVEFORM_TILE_MAX_BYTES
        while True:
            out = np.empty(cap, np.uint8)
            n, rev, hit = C.c_size_t(), C.c_uint64(), C.c_int()
            rc = lib.th_tile_cache_lookup(self.handle, track_id, ch, level, tile_index, C.byref(rev), _ptr(out, c_u8p),
                                          out.size, C.byref(n), C.byref(hit))
            if rc == _ffi.ERR_BUFFER_TOO_SMALL and n.value > cap:
                cap = n.value
                continue
            check(rc)
            return rev.value, (out[: n.value].tobytes() if hit.value else None)

    def store(self, track_id: int, ch: int, revision: int, level: int, tile_index: int, data: bytes):
        a = np.frombuffer(bytes(data), np.uint8)
        check(lib.th_tile_cache_store(self.handle, track_id, ch, revision, level, tile_index, _ptr(a, c_u8p), a.size))

    def invalidate_waveform(self):
        check(lib.th_tile_cache_invalidate(self.handle, 1, 0))

    def invalidate_spectrogram(self):
        check(lib.th_tile_cache_invalidate(self.handle, 0, 1))

    def invalidate_all(self):
        check(lib.th_tile_cache_invalidate(self.handle, 1, 1))

    def set_budget(self, budget_bytes: int):
        check(lib.th_tile_cache_set_budget(self.handle, budget_bytes))

    def stats(self) -> dict:
        e, b, bud = C.c_size_t(), C.c_size_t(), C.c_size_t()
        wr, sr, h, m = C.c_uint64(), C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.th_tile_cache_stats(self.handle, C.byref(e), C.byref(b), C.byref(bud), C.byref(wr), C.byref(sr),
                                      C.byref(h), C.byref(m)))
        return {"entries": e.value, "bytes": b.value, "budget_bytes": bud.value, "waveform_revision": wr.value,
                "spectrogram_revision": sr.value, "hits": h.value, "misses": m.value}


class TrackManager:
    """th_tm: mirror of core/mod.rs TrackManager with HBM-resident audio / specs / images."""

    def __init__(self, ctx: Context):
        h = vp()
        check(lib.th_tm_create(ctx.handle, C.byref(h)))
        self.handle, self.ctx = h, ctx
        # pinned output buffers of get_spectrogram_tiles(pinned=True): a small POOL, one buffer checked out per call (tile
        # getters run concurrently from IPC threads and ctypes drops the GIL during the C call — ADVICE r4: a buffer shared
        # by two calls in flight let one caller read another's pixels; ADVICE r5: a buffer per THREAD leaked one pinned block
        # per exited thread).  At most PIN_POOL_MAX idle buffers are kept; close() frees them all.
        self._pin_free = []  # idle buffers: (capacity, c_void_p), under _pin_lock
        self._pin_all = {}   # address -> c_void_p of every live pinned buffer, idle or checked out (for close())
        self._pin_lock = threading.Lock()

    PIN_POOL_MAX = 4

    def _pin_checkout(self, nbytes: int):
        """An idle pinned buffer of >= nbytes (the smallest that fits), else a new one.  The caller owns it until _pin_return."""
        with self._pin_lock:
            fit = [i for i, (cap, _) in enumerate(self._pin_free) if cap >= nbytes]
            if fit:
                return self._pin_free.pop(min(fit, key=lambda i: self._pin_free[i][0]))
        p = C.c_void_p()
        check(lib.th_host_alloc(self.ctx.handle, nbytes, C.byref(p)))
        with self._pin_lock:
            self._pin_all[p.value] = p
        return nbytes, p

    def _pin_return(self, cap: int, p) -> None:
        drop = None
        with self._pin_lock:
            self._pin_free.append((cap, p))
            if len(self._pin_free) > self.PIN_POOL_MAX:   # keep the large ones: they serve every request size
                drop = self._pin_free.pop(min(range(len(self._pin_free)), key=lambda i: self._pin_free[i][0]))
                self._pin_all.pop(drop[1].value, None)
        if drop is not None:
            check(lib.th_host_free(self.ctx.handle, drop[1]))

    def _free_pinned(self):
        """close(): every buffer (the caller guarantees no tile call is in flight, as for the handle itself)"""
        with self._pin_lock:
            bufs, self._pin_all, self._pin_free = list(self._pin_all.values()), {}, []
        for p in bufs:
            check(lib.th_host_free(self.ctx.handle, p))

    def close(self):
        if self.handle:
            self._free_pinned()
            check(lib.th_tm_destroy(self.handle))
            self.handle = None

    def set_colormap(self, rgba: bytes):
        a = np.frombuffer(bytes(rgba), np.uint8)
        check(lib.th_tm_set_colormap(self.handle, _ptr(a, c_u8p), a.size))

    def set_setting(self, win_ms: float, t_overlap: int, f_overlap: int, freq_scale: int):
        check(lib.th_tm_set_setting(self.handle, win_ms, t_overlap, f_overlap, freq_scale))

    def set_dB_range(self, dB_range: float):
        check(lib.th_tm_set_dB_range(self.handle, dB_range))

    def add_tracks(self, tracks):
        """tracks: iterable of (id, sr, planar ndarray [channels, samples])."""
        tracks = [(i, sr, np.atleast_2d(_f32(w))) for i, sr, w in tracks]
        n = len(tracks)
        ids = (C.c_size_t * n)(*[t[0] for t in tracks])
        srs = (C.c_uint32 * n)(*[t[1] for t in tracks])
        nch = (C.c_uint32 * n)(*[t[2].shape[0] for t in tracks])
        ns = (C.c_size_t * n)(*[t[2].shape[1] for t in tracks])
        rows = [np.ascontiguousarray(t[2][c]) for t in tracks for c in range(t[2].shape[0])]
        ptrs = (c_f32p * len(rows))(*[_ptr(r, c_f32p) for r in rows])
        check(lib.th_tm_add_tracks(self.handle, n, ids, srs, nch, ptrs, ns))

    def remove_track(self, track_id: int):
        check(lib.th_tm_remove_track(self.handle, track_id))

    def apply_track_list_changes(self):
        cap = 4096
        ids = (C.c_size_t * cap)()
        n, sr = C.c_size_t(), C.c_uint32()
        check(lib.th_tm_apply_track_list_changes(self.handle, ids, cap, C.byref(n), C.byref(sr)))
        return sorted(ids[i] for i in range(min(n.value, cap))), sr.value

    def db_state(self):
        lo, hi, sr = C.c_float(), C.c_float(), C.c_uint32()
        check(lib.th_tm_get_db_state(self.handle, C.byref(lo), C.byref(hi), C.byref(sr)))
        return lo.value, hi.value, sr.value

    def revisions(self):
        w, s = C.c_uint64(), C.c_uint64()
        check(lib.th_tm_revisions(self.handle, C.byref(w), C.byref(s)))
        return w.value, s.value

    def spec(self, track_id: int, ch: int) -> np.ndarray:
        t, h = C.c_size_t(), C.c_size_t()
        check(lib.th_tm_spec_shape(self.handle, track_id, ch, C.byref(t), C.byref(h)))
        out = np.empty((t.value, h.value), np.float32)
        check(lib.th_tm_copy_spec(self.handle, track_id, ch, _ptr(out, c_f32p), out.size))
        return out

    def img(self, track_id: int, ch: int) -> np.ndarray:
        h, w = C.c_size_t(), C.c_size_t()
        check(lib.th_tm_img_shape(self.handle, track_id, ch, C.byref(h), C.byref(w)))
        out = np.empty((h.value, w.value), np.uint16)
        check(lib.th_tm_copy_img(self.handle, track_id, ch, _ptr(out, c_u16p), out.size))
        return out

    def get_spectrogram_tile(self, track_id: int, ch: int, level_x: int, level_y: int, tile_x: int,
                             tile_y: int) -> bytes:
        out = np.empty(SPECTROGRAM_TILE_MAX_BYTES, np.uint8)
        n = C.c_size_t()
        check(lib.th_tm_get_spectrogram_tile(self.handle, track_id, ch, level_x, level_y, tile_x, tile_y,
                                             _ptr(out, c_u8p), out.size, C.byref(n)))
        return out[: n.value].tobytes()

    def get_spectrogram_tiles(self, requests, pinned: bool = False):
        """th_tm_get_spectrogram_tiles: requests = iterable of (track_id, ch, level_x, level_y, tile_x, tile_y); returns the
        list of tile byte strings (each equal to get_spectrogram_tile's).  pinned: the output buffer is pinned host memory
        (th_host_alloc), which the raster kernel writes directly; else a pageable numpy buffer (staged + copied)."""
        reqs = list(requests)
        n = len(reqs)
        arr = (_ffi.TileRequest * max(n, 1))(*[_ffi.TileRequest(*r, 0) for r in reqs])
        offs = (C.c_size_t * (n + 1))()
        need = C.c_size_t()
        rc = lib.th_tm_get_spectrogram_tiles(self.handle, arr, n, None, 0, offs, C.byref(need))
        if rc not in (_ffi.OK, _ffi.ERR_BUFFER_TOO_SMALL):
            check(rc)
        if n == 0 or need.value == 0:
            return []
        if pinned:
            # a pooled pinned buffer for the duration of this call (a hipHostMalloc / hipHostFree pair per call
            # costs milliseconds, most of what the direct write saves; ADVICE r4: never shared by two calls in flight)
            cap, pin = self._pin_checkout(need.value)
            try:
                check(lib.th_tm_get_spectrogram_tiles(self.handle, arr, n, pin, cap, offs, C.byref(need)))
                out = bytes((C.c_uint8 * need.value).from_address(pin.value))
            finally:
                self._pin_return(cap, pin)
        else:
            b = np.empty(need.value, np.uint8)
            check(lib.th_tm_get_spectrogram_tiles(self.handle, arr, n, b.ctypes.data_as(C.c_void_p), b.size, offs, C.byref(need)))
            out = b.tobytes()
        tiles = []
        for i in range(n):
            w, h = struct.unpack_from("<II", out, offs[i] + 8)
            tiles.append(out[offs[i]: offs[i] + 40 + 4 * w * h])
        return tiles

    def set_lod_source(self, per_request: bool) -> None:
        """LOD > 0 tiles from the resident mip pyramid (False, default) or resampled per request (True)."""
        check(lib.th_tm_set_lod_source(self.handle, int(per_request)))

    def mip_level(self, track_id: int, ch: int, level_x: int, level_y: int) -> np.ndarray:
        """Dense copy of one resident level of the channel's LOD mip pyramid ((0, 0): the image itself)."""
        w, h = C.c_size_t(), C.c_size_t()
        check(lib.th_tm_mip_level(self.handle, track_id, ch, level_x, level_y, None, 0, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value), np.uint16)
        check(lib.th_tm_mip_level(self.handle, track_id, ch, level_x, level_y, _ptr(out, c_u16p), out.size, C.byref(w),
                                  C.byref(h)))
        return out

    def put_img(self, track_id: int, ch: int, img: np.ndarray) -> None:
        """Replace the pixels of a resident u16 image (same shape) and rebuild its mip pyramid (th_tm_put_img)."""
        img = np.ascontiguousarray(img, dtype=np.uint16)
        check(lib.th_tm_put_img(self.handle, track_id, ch, _ptr(img, c_u16p), img.shape[0], img.shape[1]))

    def lod_footprint(self) -> dict:
        """device memory of the LOD machinery: tap tables (count, bytes) and mip pyramids (bytes)"""
        n, ab, mb = C.c_size_t(), C.c_size_t(), C.c_size_t()
        check(lib.th_tm_lod_footprint(self.handle, C.byref(n), C.byref(ab), C.byref(mb)))
        return {"axis_tables": n.value, "axis_table_bytes": ab.value, "mip_bytes": mb.value}

    def render_metadata(self, track_id: int, ch: int, track_sec: float, is_clipped: bool) -> dict:
        """AudioRenderMetadata of get_audio_render_metadata (lib.rs:321-340)."""
        m = _ffi.RenderMetadata()
        check(lib.th_tm_get_audio_render_metadata(self.handle, track_id, ch, track_sec, int(is_clipped), C.byref(m)))
        return {k: getattr(m, k) for k, _ in m._fields_}

    def tile_cache(self) -> TileCache:
        """The RenderTileCache in front of get_waveform_tile (borrowed; owned by the manager)."""
        h = vp()
        check(lib.th_tm_tile_cache(self.handle, C.byref(h)))
        return TileCache(_borrowed=h)

    def get_waveform_tile(self, track_id: int, ch: int, level: int, tile_index: int) -> bytes:
        out = np.empty(WAVEFORM_TILE_MAX_BYTES, np.uint8)
        n = C.c_size_t()
        check(lib.th_tm_get_waveform_tile(self.handle, track_id, ch, level, tile_index, _ptr(out, c_u8p), out.size,
                                          C.byref(n)))
        return out[: n.value].tobytes()
