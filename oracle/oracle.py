"""ctypes binding of oracle/thesia_oracle.c — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module, and only as the checker / the reported CPU baseline.  The product
package (thesia_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libthesia_oracle.so")

LINEAR, MEL = 0, 1


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "thesia_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_u16p = C.POINTER(C.c_uint16)
_u8p = C.POINTER(C.c_uint8)


class _TileGeom(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("origin_x", C.c_uint32), ("origin_y", C.c_uint32),
                ("lod_w", C.c_size_t), ("lod_h", C.c_size_t)]


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_stft_n_frames.restype = C.c_size_t
        L.orc_stft_n_frames.argtypes = [C.c_size_t] * 3
        L.orc_perform_stft.restype = C.c_size_t
        L.orc_perform_stft.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, C.c_int, _f32p]
        L.orc_calc_spec.restype = C.c_size_t
        L.orc_calc_spec.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _f32p, _f32p,
                                    C.c_size_t, C.c_int, _f32p, _f32p]
        L.orc_calc_framing_params.argtypes = [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_hann.argtypes = [C.c_size_t, C.c_int, _f32p]
        L.orc_calc_normalized_win.argtypes = [C.c_size_t, C.c_size_t, _f32p]
        L.orc_pad_reflect.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, _f32p]
        L.orc_pad_constant.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, _f32p]
        L.orc_dB_from_amp_inplace.argtypes = [_f32p, C.c_size_t, C.c_float, C.c_float]
        for n in ("orc_mel_to_hz_f64", "orc_mel_from_hz_f64"):
            getattr(L, n).restype = C.c_double
            getattr(L, n).argtypes = [C.c_double]
        for n in ("orc_mel_to_hz_f32", "orc_mel_from_hz_f32"):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float]
        L.orc_calc_mel_fb_f32.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, _f32p]
        L.orc_calc_mel_fb_f64.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, C.c_double, C.c_double, C.c_int, _f64p]
        L.orc_mel_default_n_mel.restype = C.c_size_t
        L.orc_mel_default_n_mel.argtypes = [C.c_uint32, C.c_size_t]
        L.orc_hz_range_to_idx.argtypes = [C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_size_t,
                                          C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
        L.orc_find_min_max.argtypes = [_f32p, C.c_size_t, _f32p, _f32p]
        L.orc_sum_avx2.restype = C.c_float
        L.orc_sum_squares.restype = C.c_float
        L.orc_sum_squares.argtypes = [_f32p, C.c_size_t]
        L.orc_abs_max.restype = C.c_float
        L.orc_abs_max.argtypes = [_f32p, C.c_size_t]
        L.orc_sum_avx2.argtypes = [_f32p, C.c_size_t, C.c_size_t]
        L.orc_scalar_mul.argtypes = [_f32p, C.c_size_t, C.c_float]
        L.orc_global_db_range.argtypes = [_f32p, _f32p, C.c_size_t, C.c_float, _f32p, _f32p]
        L.orc_convert_spectrogram_to_img.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t,
                                                     C.c_float, C.c_float, C.c_uint32, _u16p]
        L.orc_encode_waveform_tile.restype = C.c_size_t
        L.orc_encode_waveform_tile.argtypes = [_f32p, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, _u8p]
        L.orc_encode_spectrogram_tile.restype = C.c_size_t
        L.orc_encode_spectrogram_tile.argtypes = [_u16p, C.c_size_t, C.c_size_t, _u8p, C.c_size_t, C.c_uint64,
                                                  C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _u8p]
        L.orc_resize_whole_image.argtypes = [_u16p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_int, _u16p]
        L.orc_spectrogram_tile_u16.restype = C.c_size_t
        L.orc_spectrogram_tile_u16.argtypes = [_u16p, C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.c_uint32, C.c_int, _u16p, C.POINTER(_TileGeom)]
        L.orc_track_step.restype = C.c_size_t
        L.orc_track_step.argtypes = [_f32p, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, _u8p, C.c_size_t, C.c_float,
                                     C.POINTER(C.c_uint64)]
        _lib = L
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a, t):
    return a.ctypes.data_as(t)


# ---- a1 ----
def calc_framing_params(win_ms: float, t_overlap: int, f_overlap: int, sr: int):
    h, w, n = C.c_size_t(), C.c_size_t(), C.c_size_t()
    lib().orc_calc_framing_params(win_ms, t_overlap, f_overlap, sr, C.byref(h), C.byref(w), C.byref(n))
    return h.value, w.value, n.value


# ---- a2 ----
def hann(size: int, symmetric: bool = False) -> np.ndarray:
    out = np.empty(size, np.float32)
    lib().orc_hann(size, int(symmetric), _p(out, _f32p))
    return out


def calc_normalized_win(win: int, n_fft: int) -> np.ndarray:
    out = np.empty(win, np.float32)
    lib().orc_calc_normalized_win(win, n_fft, _p(out, _f32p))
    return out


# ---- a3 ----
def pad_reflect(x, left: int, right: int) -> np.ndarray:
    x = _f32(x)
    out = np.empty(len(x) + left + right, np.float32)
    lib().orc_pad_reflect(_p(x, _f32p), len(x), left, right, _p(out, _f32p))
    return out


def pad_constant(x, left: int, right: int, c: float) -> np.ndarray:
    x = _f32(x)
    out = np.empty(len(x) + left + right, np.float32)
    lib().orc_pad_constant(_p(x, _f32p), len(x), left, right, c, _p(out, _f32p))
    return out


# ---- a4/a5 ----
def stft_n_frames(n: int, win: int, hop: int) -> int:
    return lib().orc_stft_n_frames(n, win, hop)


def perform_stft(x, win: int, hop: int, n_fft: int, window=None, fft32: bool = False) -> np.ndarray:
    x = _f32(x)
    if window is None:
        window = calc_normalized_win(win, n_fft)
    window = _f32(window)
    T = stft_n_frames(len(x), win, hop)
    F = n_fft // 2 + 1
    out = np.zeros((T, F, 2), np.float32)
    T2 = lib().orc_perform_stft(_p(x, _f32p), len(x), win, hop, n_fft, _p(window, _f32p), int(fft32),
                                _p(out, _f32p))
    assert T2 == T, (T2, T)
    return out[..., 0] + 1j * out[..., 1]


def calc_spec(x, win: int, hop: int, n_fft: int, mel_fb=None, fft32: bool = False, return_amp: bool = False):
    """calc_spec (spectrogram.rs:187-212): T x H f32 dB."""
    x = _f32(x)
    window = calc_normalized_win(win, n_fft)
    T = stft_n_frames(len(x), win, hop)
    F = n_fft // 2 + 1
    n_mel = 0
    fbp = None
    if mel_fb is not None:
        mel_fb = _f32(mel_fb)
        assert mel_fb.shape[0] == F
        n_mel = mel_fb.shape[1]
        fbp = _p(mel_fb, _f32p)
    H = n_mel if n_mel else F
    out = np.empty((T, H), np.float32)
    amp = np.empty((T, F), np.float32) if return_amp else None
    T2 = lib().orc_calc_spec(_p(x, _f32p), len(x), win, hop, n_fft, _p(window, _f32p), fbp, n_mel, int(fft32),
                             _p(out, _f32p), _p(amp, _f32p) if return_amp else None)
    assert T2 == T, (T2, T)
    return (out, amp) if return_amp else out


# ---- a9 ----
def dB_from_amp(x, ref_value: float = 1.0, amin: float = 0.0) -> np.ndarray:
    y = _f32(x).copy().ravel()
    lib().orc_dB_from_amp_inplace(_p(y, _f32p), y.size, ref_value, amin)
    return y.reshape(np.shape(x))


# ---- a7 ----
def mel_to_hz(m, f32=False):
    return lib().orc_mel_to_hz_f32(m) if f32 else lib().orc_mel_to_hz_f64(m)


def mel_from_hz(h, f32=False):
    return lib().orc_mel_from_hz_f32(h) if f32 else lib().orc_mel_from_hz_f64(h)


def calc_mel_fb(sr: int, n_fft: int, n_mel: int, fmin: float = 0.0, fmax=None, do_norm: bool = True,
                dtype=np.float32) -> np.ndarray:
    F = n_fft // 2 + 1
    out = np.empty((F, n_mel), dtype)
    fm = -1.0 if fmax is None else fmax
    if dtype == np.float32:
        lib().orc_calc_mel_fb_f32(sr, n_fft, n_mel, fmin, fm, int(do_norm), _p(out, _f32p))
    else:
        lib().orc_calc_mel_fb_f64(sr, n_fft, n_mel, fmin, fm, int(do_norm), _p(out, _f64p))
    return out


def mel_fb_points(sr: int, n_fft: int, n_mel: int, fmin: float = 0.0, fmax=None):
    """(lin, mf): the f32 bin frequencies and triangle points calc_mel_fb works from (lib.rs:61-67)"""
    lin, mf = np.empty(n_fft // 2 + 1, np.float32), np.empty(n_mel + 2, np.float32)
    L = lib()
    L.orc_mel_fb_points_f32.restype = None
    L.orc_mel_fb_points_f32.argtypes = [C.c_uint32, C.c_size_t, C.c_size_t, C.c_float, C.c_float, _f32p, _f32p]
    L.orc_mel_fb_points_f32(sr, n_fft, n_mel, fmin, -1.0 if fmax is None else fmax, _p(lin, _f32p), _p(mf, _f32p))
    return lin, mf


def mel_default_n_mel(sr: int, n_fft: int) -> int:
    return lib().orc_mel_default_n_mel(sr, n_fft)


def calc_mel_fb_default(sr: int, n_fft: int) -> np.ndarray:
    return calc_mel_fb(sr, n_fft, mel_default_n_mel(sr, n_fft))


def hz_range_to_idx(freq_scale: int, hz_range, sr: int, n: int):
    a, b = C.c_size_t(), C.c_size_t()
    lib().orc_hz_range_to_idx(freq_scale, hz_range[0], hz_range[1], sr, n, C.byref(a), C.byref(b))
    return a.value, b.value


# ---- a10 ----
def find_min_max(x):
    x = _f32(x).ravel()
    mn, mx = C.c_float(), C.c_float()
    lib().orc_find_min_max(_p(x, _f32p), x.size, C.byref(mn), C.byref(mx))
    return mn.value, mx.value


def sum_avx2(x, misalign: int = 0) -> float:
    x = _f32(x).ravel()
    return lib().orc_sum_avx2(_p(x, _f32p), x.size, misalign)


def sum_squares(x) -> float:
    """simd.rs:820-832 (Kahan, scalar tier)"""
    x = _f32(x).ravel()
    return lib().orc_sum_squares(_p(x, _f32p), x.size)


def abs_max(x) -> float:
    """simd.rs:935-937"""
    x = _f32(x).ravel()
    return lib().orc_abs_max(_p(x, _f32p), x.size)


def scalar_mul(x, s: float) -> np.ndarray:
    y = _f32(x).copy().ravel()
    lib().orc_scalar_mul(_p(y, _f32p), y.size, s)
    return y


def global_db_range(mins, maxs, dB_range: float = 100.0):
    mins, maxs = _f32(mins).ravel(), _f32(maxs).ravel()
    mn, mx = C.c_float(), C.c_float()
    lib().orc_global_db_range(_p(mins, _f32p), _p(maxs, _f32p), mins.size, dB_range, C.byref(mn), C.byref(mx))
    return mn.value, mx.value


# ---- a12 ----
def convert_spectrogram_to_img(spec, i_freq_range, dB_range, colormap_length=None) -> np.ndarray:
    spec = _f32(spec)
    T, H = spec.shape
    i0, i1 = i_freq_range
    out = np.empty((i1 - i0, T), np.uint16)
    lib().orc_convert_spectrogram_to_img(_p(spec, _f32p), T, H, i0, i1, dB_range[0], dB_range[1],
                                         0 if colormap_length is None else colormap_length, _p(out, _u16p))
    return out


# ---- a14 ----
def encode_waveform_tile(wav, revision: int, level: int, tile_index: int) -> bytes:
    wav = _f32(wav)
    out = np.empty(24 + 1024 * 12, np.uint8)
    n = lib().orc_encode_waveform_tile(_p(wav, _f32p), wav.size, revision, level, tile_index, _p(out, _u8p))
    return out[:n].tobytes()


# ---- a13 ----
def encode_spectrogram_tile(img, colormap_rgba, revision: int, level_x: int, level_y: int, tile_x: int,
                            tile_y: int) -> bytes:
    img = np.ascontiguousarray(img, dtype=np.uint16)
    cm = np.ascontiguousarray(np.frombuffer(bytes(colormap_rgba), np.uint8))
    Hh, W = img.shape
    out = np.empty(40 + 520 * 520 * 4, np.uint8)
    n = lib().orc_encode_spectrogram_tile(_p(img, _u16p), Hh, W, _p(cm, _u8p), cm.size, revision, level_x,
                                          level_y, tile_x, tile_y, _p(out, _u8p))
    return out[:n].tobytes()


# LOD resize variants (sensitivity checks; see the C file): mode -1 = the tile encoder's own f64 Lanczos3,
# 0 = f64 with pre-normalised taps, 1 = fixed point at the largest precision an i32 coefficient allows, 2 = 16-bit fixed point
RESIZE_F64, RESIZE_F64_PRENORM, RESIZE_FIXED_MAX, RESIZE_FIXED_16 = -1, 0, 1, 2


def resize_whole_image(img, level_x: int, level_y: int, mode: int = RESIZE_F64) -> np.ndarray:
    """The whole image resampled to LOD (level_x, level_y): what a pre-built mip level holds (rows low -> high frequency)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    Hh, W = img.shape
    out = np.empty((-(-Hh // (1 << level_y)), -(-W // (1 << level_x))), np.uint16)
    lib().orc_resize_whole_image(_p(img, _u16p), Hh, W, level_x, level_y, mode, _p(out, _u16p))
    return out


def spectrogram_tile_u16(img, level_x: int, level_y: int, tile_x: int, tile_y: int, mode: int = RESIZE_F64):
    """LOD pixels of one tile before the row flip and the colour map (render_tiles.rs:330-338) + (origin_x, origin_y)."""
    img = np.ascontiguousarray(img, dtype=np.uint16)
    Hh, W = img.shape
    out = np.empty(520 * 520, np.uint16)
    g = _TileGeom()
    n = lib().orc_spectrogram_tile_u16(_p(img, _u16p), Hh, W, level_x, level_y, tile_x, tile_y, mode, _p(out, _u16p),
                                       C.byref(g))
    assert n == g.width * g.height
    return out[:n].reshape(g.height, g.width).copy(), (g.origin_x, g.origin_y)


def track_step(x, win: int, hop: int, n_fft: int, colormap_rgba: bytes, dB_range: float = 100.0) -> int:
    """One track through the whole benchmark step on the CPU (cpu_baseline leg); returns #frames."""
    x = _f32(x)
    cm = np.frombuffer(bytes(colormap_rgba), np.uint8)
    chk = C.c_uint64()
    return lib().orc_track_step(_p(x, _f32p), x.size, win, hop, n_fft, _p(cm, _u8p), cm.size, dB_range, C.byref(chk))


# ---------------------------------------------------------------------------------------------
# RenderTileCache — src-tauri/src/core/render_tiles.rs:51-230 (waveform-tile LRU + revisions).
# Literal restatement: a dict of entries with last_used ticks and the reference's linear min_by_key
# eviction scan.  Pinned by the reference's four cache tests (render_tiles.rs:473-537), restated in
# tests/test_tile_cache.py.
# ---------------------------------------------------------------------------------------------
class RenderTileCache:
    DEFAULT_BUDGET = 32 * 1024 * 1024  # render_tiles.rs:17

    def __init__(self, budget_bytes: int = DEFAULT_BUDGET):  # with_budget, :68-78
        self.entries = {}  # key -> [bytes, last_used]
        self.bytes = 0
        self.budget_bytes = budget_bytes
        self.tick = 0
        self.waveform_revision = 1
        self.spectrogram_revision = 1

    def _next_tick(self) -> int:  # :226-228
        self.tick = (self.tick + 1) & 0xFFFFFFFFFFFFFFFF
        return self.tick

    def invalidate_waveform(self):  # :87-90
        self.waveform_revision = max((self.waveform_revision + 1) & 0xFFFFFFFFFFFFFFFF, 1)
        self.entries.clear()  # clear_tiles, :220-224
        self.bytes = 0

    def invalidate_spectrogram(self):  # :92-94
        self.spectrogram_revision = max((self.spectrogram_revision + 1) & 0xFFFFFFFFFFFFFFFF, 1)

    def invalidate_all(self):  # :96-99
        self.invalidate_waveform()
        self.invalidate_spectrogram()

    def cached_waveform_tile(self, id_, ch, level, tile_index):  # :124-144
        revision = self.waveform_revision
        e = self.entries.get((id_, ch, revision, level, tile_index))
        if e is None:
            return revision, None
        e[1] = self._next_tick()
        return revision, e[0]

    def store_waveform_tile(self, id_, ch, revision, level, tile_index, data: bytes):  # :146-169
        if revision != self.waveform_revision:
            return
        self._insert((id_, ch, revision, level, tile_index), bytes(data))

    def _insert(self, key, data):  # :190-203
        old = self.entries.get(key)
        self.entries[key] = [data, self._next_tick()]
        if old is not None:
            self.bytes = max(self.bytes - len(old[0]), 0)
        self.bytes += len(data)
        self._evict()

    def _evict(self):  # :205-218
        while self.bytes > self.budget_bytes:
            if not self.entries:
                break
            key = min(self.entries, key=lambda k: self.entries[k][1])
            self.bytes -= len(self.entries.pop(key)[0])
