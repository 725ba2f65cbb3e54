"""ctypes binding of libthesia_amd.so (the C ABI declared in include/thesia_amd.h).

The shared library is built in-tree by ``__graft_entry__.build()`` (hipcc, gfx950).  Importing
this module fails loudly when it is missing: there is no CPU fallback of any kind.
"""
from __future__ import annotations

import ctypes as C
import importlib.util
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# THESIA_AMD_LIB: development override to A/B a variant build of the same library (scripts/build_variant.sh)
LIB_PATH = os.environ.get("THESIA_AMD_LIB") or os.path.join(_HERE, "libthesia_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} not found: build the HIP extension first "
        "(python -c 'import __graft_entry__ as g; g.build()').  thesia_amd has no CPU fallback."
    )



def _preload_hip_runtime() -> None:
    """One HIP runtime per process.  libthesia_amd.so needs `libamdhip64.so.7` by soname; PyTorch
    wheels bundle their own copy under torch/lib with that same soname.  If torch is (or may later
    be) in the process, load torch's copy first so the dynamic loader resolves our NEEDED entry to
    it instead of mapping /opt/rocm's as a second runtime (two runtimes = "no device" in the second).
    THESIA_AMD_SYSTEM_HIP=1 keeps the system runtime (pure C hosts, no torch)."""
    if "torch" in sys.modules or os.environ.get("THESIA_AMD_SYSTEM_HIP") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is not None and spec.origin:
        cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)


_preload_hip_runtime()
lib = C.CDLL(LIB_PATH)

c_f32p = C.POINTER(C.c_float)
c_u8p = C.POINTER(C.c_uint8)
c_u16p = C.POINTER(C.c_uint16)
c_szp = C.POINTER(C.c_size_t)
vp = C.c_void_p


class ThError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"thesia_amd error {code}: {msg}")
        self.code = code


OK, ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_HIP, ERR_NO_DEVICE, ERR_OOM, ERR_BUFFER_TOO_SMALL, ERR_NOT_FOUND, \
    ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6, -7, -8


class TileRequest(C.Structure):
    _fields_ = [("id", C.c_size_t), ("ch", C.c_uint32), ("level_x", C.c_uint32), ("level_y", C.c_uint32),
                ("tile_x", C.c_uint32), ("tile_y", C.c_uint32), ("reserved", C.c_uint32)]


class StatsDesc(C.Structure):
    _fields_ = [("wav", C.c_void_p), ("n_samples", C.c_uint64)]


class PyramidDesc(C.Structure):
    _fields_ = [("wav", C.c_void_p), ("out", C.c_void_p), ("n_samples", C.c_uint64), ("n_levels", C.c_uint32),
                ("first_level", C.c_uint32)]


class RenderMetadata(C.Structure):
    _fields_ = [("waveform_revision", C.c_uint64), ("spectrogram_revision", C.c_uint64), ("sample_rate", C.c_uint32),
                ("is_clipped", C.c_uint32), ("sample_count", C.c_uint64), ("track_sec", C.c_double),
                ("spectrogram_width", C.c_uint64), ("spectrogram_height", C.c_uint64), ("waveform_tile_bins", C.c_uint64),
                ("spectrogram_tile_size", C.c_uint64)]


class TileGeom(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("origin_x", C.c_uint32), ("origin_y", C.c_uint32),
                ("lod_width", C.c_uint64), ("lod_height", C.c_uint64)]


class ChanDesc(C.Structure):
    _fields_ = [("wav", vp), ("spec", vp), ("n_samples", C.c_uint64), ("n_frames", C.c_uint64),
                ("spec_pitch", C.c_uint64)]


class ImgDesc(C.Structure):
    _fields_ = [("spec", vp), ("img", vp), ("n_frames", C.c_uint64), ("height", C.c_uint64),
                ("i_start", C.c_uint64), ("i_end", C.c_uint64), ("spec_pitch", C.c_uint64), ("img_pitch", C.c_uint64)]


class RasterDesc(C.Structure):
    _fields_ = [("img", vp), ("rgba", vp), ("img_width", C.c_uint32), ("img_height", C.c_uint32),
                ("origin_x", C.c_uint32), ("origin_y", C.c_uint32), ("width", C.c_uint32), ("height", C.c_uint32),
                ("img_pitch", C.c_uint32), ("reserved", C.c_uint32)]


class ImgTilesDesc(C.Structure):  # th_img_tiles_desc
    _fields_ = [("img", ImgDesc), ("tiles", C.POINTER(vp)), ("n_tiles_x", C.c_uint32), ("n_tiles_y", C.c_uint32)]


class WaveDesc(C.Structure):
    _fields_ = [("wav", vp), ("bins", vp), ("n_samples", C.c_uint64), ("start", C.c_uint64),
                ("level", C.c_uint32), ("bin_count", C.c_uint32)]


# name -> argtypes (restype is int unless listed in _RESTYPES).  Keep in sync with include/thesia_amd.h;
# tests/test_abi.py checks that every TH_API symbol of the header is exported and declared here.
_SIGS = {
    "th_version": [],
    "th_device_count": [C.POINTER(C.c_int)],
    "th_calc_framing_params": [C.c_double, C.c_uint32, C.c_uint32, C.c_uint32, c_szp, c_szp, c_szp],
    "th_stft_n_frames": [C.c_size_t, C.c_size_t, C.c_size_t, c_szp],
    "th_calc_normalized_win": [C.c_size_t, C.c_size_t, c_f32p],
    "th_calc_mel_fb": [C.c_uint32, C.c_size_t, C.c_size_t, C.c_float, C.c_float, C.c_int, c_f32p],
    "th_mel_default_n_mel": [C.c_uint32, C.c_size_t, c_szp],
    "th_hz_range_to_idx": [C.c_int, C.c_float, C.c_float, C.c_uint32, C.c_size_t, c_szp, c_szp],
    "th_global_db_range": [c_f32p, c_f32p, C.c_size_t, C.c_float, c_f32p, c_f32p],
    "th_shard_assign": [C.POINTER(C.c_uint64), C.c_size_t, C.c_uint32, C.POINTER(C.c_uint32)],
    "th_spectrogram_tile_geometry": [C.c_size_t, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                     C.POINTER(TileGeom)],
    "th_waveform_tile_geometry": [C.c_size_t, C.c_uint32, C.c_uint32, c_szp, c_szp, c_szp],
    "th_ctx_create": [C.c_int, vp, C.POINTER(vp)],
    "th_ctx_create_ex": [C.c_int, vp, C.c_int, C.POINTER(vp)],
    "th_ctx_destroy": [vp],
    "th_ctx_synchronize": [vp],
    "th_ctx_capture_begin": [vp],
    "th_ctx_capture_end": [vp, C.POINTER(vp)],
    "th_graph_launch": [vp],
    "th_graph_destroy": [vp],
    "th_dev_alloc": [vp, C.c_size_t, C.POINTER(vp)],
    "th_dev_free": [vp, vp],
    "th_dev_upload": [vp, vp, vp, C.c_size_t],
    "th_dev_download": [vp, vp, vp, C.c_size_t],
    "th_dev_copy": [vp, vp, vp, C.c_size_t],
    "th_timer_start": [vp],
    "th_timer_stop_ms": [vp, c_f32p],
    "th_plan_create": [vp, C.c_uint32, C.c_size_t, C.c_size_t, C.c_size_t, C.c_int, C.c_size_t, C.POINTER(vp)],
    "th_plan_destroy": [vp],
    "th_plan_dims": [vp, c_szp, c_szp],
    "th_plan_set_kernel": [vp, C.c_int],
    "th_build_ab_variants": [],
    "th_plan_mel_moments_info": [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_double), C.POINTER(C.c_double)],
    "th_plan_kernel_name": [vp],
    "th_calc_spec_batch_dev": [vp, C.POINTER(ChanDesc), C.c_size_t, vp],
    "th_calc_spec_batch_ranged_dev": [vp, C.POINTER(ChanDesc), C.c_size_t, vp, C.c_float, vp],
    "th_calc_spec_host": [vp, c_f32p, C.c_size_t, c_f32p, C.c_size_t, c_szp, c_f32p, c_f32p],
    "th_spec_to_img_dev": [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, C.c_size_t, C.c_float, C.c_float,
                           C.c_uint32, vp],
    "th_spec_to_img_batch_dev": [vp, C.POINTER(ImgDesc), C.c_size_t, C.c_float, C.c_float, C.c_uint32],
    "th_pitch_f32": [C.c_size_t],
    "th_pitch_u16": [C.c_size_t],
    "th_encode_spectrogram_tile_dev": [vp, vp, C.c_size_t, C.c_size_t, C.c_size_t, c_u8p, C.c_size_t, C.c_uint64, C.c_uint32,
                                       C.c_uint32, C.c_uint32, C.c_uint32, c_u8p, C.c_size_t, c_szp],
    "th_raster_tiles_dev": [vp, C.POINTER(RasterDesc), C.c_size_t, vp, C.c_uint32],
    "th_spec_to_img_raster_batch_dev": [vp, C.POINTER(ImgTilesDesc), C.c_size_t, C.c_float, C.c_float, C.c_void_p, vp, C.c_uint32],
    "th_encode_waveform_tile_dev": [vp, vp, C.c_size_t, C.c_uint64, C.c_uint32, C.c_uint32, c_u8p, C.c_size_t,
                                    c_szp],
    "th_waveform_tiles_dev": [vp, C.POINTER(WaveDesc), C.c_size_t],
    "th_tm_create": [vp, C.POINTER(vp)],
    "th_tm_destroy": [vp],
    "th_tm_set_colormap": [vp, c_u8p, C.c_size_t],
    "th_tm_set_setting": [vp, C.c_double, C.c_uint32, C.c_uint32, C.c_int],
    "th_tm_set_dB_range": [vp, C.c_float],
    "th_tm_add_track": [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.POINTER(c_f32p), C.c_size_t],
    "th_tm_add_tracks": [vp, C.c_size_t, c_szp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(c_f32p),
                         c_szp],
    "th_tm_remove_track": [vp, C.c_size_t],
    "th_tm_apply_track_list_changes": [vp, c_szp, C.c_size_t, c_szp, C.POINTER(C.c_uint32)],
    "th_tm_get_db_state": [vp, c_f32p, c_f32p, C.POINTER(C.c_uint32)],
    "th_tm_spec_shape": [vp, C.c_size_t, C.c_uint32, c_szp, c_szp],
    "th_tm_img_shape": [vp, C.c_size_t, C.c_uint32, c_szp, c_szp],
    "th_tm_copy_spec": [vp, C.c_size_t, C.c_uint32, c_f32p, C.c_size_t],
    "th_tm_copy_img": [vp, C.c_size_t, C.c_uint32, c_u16p, C.c_size_t],
    "th_tm_revisions": [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)],
    "th_tm_get_spectrogram_tile": [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32,
                                   c_u8p, C.c_size_t, c_szp],
    "th_tm_get_waveform_tile": [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, c_u8p, C.c_size_t, c_szp],
    "th_tm_get_spectrogram_tiles": [vp, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, c_szp, c_szp],
    "th_host_alloc": [vp, C.c_size_t, C.POINTER(C.c_void_p)],
    "th_host_free": [vp, C.c_void_p],
    "th_minmax_reduce_dev": [vp, C.c_void_p, C.c_size_t, C.c_void_p],
    "th_global_db_range_dev": [vp, C.c_void_p, C.c_float, C.c_void_p],
    "th_minmax_reduce_range_dev": [vp, vp, C.c_size_t, C.c_float, vp, vp],
    "th_spec_to_img_batch_dev_ranged": [vp, C.POINTER(ImgDesc), C.c_size_t, C.c_void_p, C.c_uint32],
    "th_plan_time_kernel": [vp, C.c_int],
    "th_plan_last_kernel_ms": [vp, C.POINTER(C.c_float)],
    "th_plan_kernel_ms_history": [vp, c_f32p, C.c_size_t, c_szp],
    "th_channel_stats_dev": [vp, C.POINTER(StatsDesc), C.c_size_t, c_f32p, c_f32p],
    "th_waveform_pyramid_bins": [C.c_uint64, C.c_uint32],
    "th_waveform_pyramid_offset": [C.c_uint64, C.c_uint32],
    "th_waveform_pyramid_dev": [vp, C.POINTER(PyramidDesc), C.c_size_t],
    "th_tm_tile_cache": [vp, C.POINTER(vp)],
    "th_tm_set_lod_source": [vp, C.c_int],
    "th_tm_mip_level": [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, c_u16p, C.c_size_t, c_szp, c_szp],
    "th_tm_put_img": [vp, C.c_size_t, C.c_uint32, c_u16p, C.c_size_t, C.c_size_t],
    "th_tm_lod_footprint": [vp, c_szp, c_szp, c_szp],
    "th_tm_get_audio_render_metadata": [vp, C.c_size_t, C.c_uint32, C.c_double, C.c_int, C.POINTER(RenderMetadata)],
    "th_tile_cache_create": [C.c_size_t, C.POINTER(vp)],
    "th_tile_cache_destroy": [vp],
    "th_tile_cache_lookup": [vp, C.c_size_t, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64), c_u8p, C.c_size_t,
                             c_szp, C.POINTER(C.c_int)],
    "th_tile_cache_store": [vp, C.c_size_t, C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, c_u8p, C.c_size_t],
    "th_tile_cache_invalidate": [vp, C.c_int, C.c_int],
    "th_tile_cache_set_budget": [vp, C.c_size_t],
    "th_tile_cache_stats": [vp, c_szp, c_szp, c_szp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64),
                            C.POINTER(C.c_uint64)],
}
_RESTYPES = {"th_plan_kernel_name": C.c_char_p, "th_pitch_f32": C.c_size_t, "th_pitch_u16": C.c_size_t,
             "th_waveform_pyramid_bins": C.c_size_t, "th_waveform_pyramid_offset": C.c_size_t}

lib.th_last_error.restype = C.c_char_p
lib.th_last_error.argtypes = []
for _name, _args in _SIGS.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch: fail loudly
    _fn.argtypes = _args
    _fn.restype = _RESTYPES.get(_name, C.c_int)


def last_error() -> str:
    return (lib.th_last_error() or b"").decode()


def check(code: int) -> None:
    if code != OK:
        raise ThError(code, last_error())
