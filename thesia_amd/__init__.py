"""thesia_amd — MI355X-native spectrogram / waveform compute path for the thesia viewer.

The product is libthesia_amd.so (hand-written HIP kernels for gfx950 behind the C ABI in
include/thesia_amd.h).  This package is the ctypes face of that ABI; importing it fails loudly
if the library has not been built.  There is no CPU fallback.
"""
from ._ffi import LIB_PATH, ThError, ChanDesc, ImgDesc, RasterDesc, WaveDesc, TileGeom  # noqa: F401
from .api import (LINEAR, MEL, Context, ab_variants, DeviceBuffer, Graph, Plan, TileCache, TrackManager, calc_framing_params,  # noqa: F401
                  calc_mel_fb, calc_normalized_win, device_count, global_db_range, hz_range_to_idx,
                  mel_default_n_mel, pitch_f32, pitch_u16, shard_assign, spectrogram_tile_geometry, stft_n_frames, waveform_tile_geometry)

__all__ = ["LINEAR", "MEL", "Context", "ab_variants", "DeviceBuffer", "Graph", "Plan", "TileCache", "TrackManager", "ThError", "calc_framing_params",
           "calc_mel_fb", "calc_normalized_win", "device_count", "global_db_range", "hz_range_to_idx",
           "mel_default_n_mel", "pitch_f32", "pitch_u16", "shard_assign", "spectrogram_tile_geometry", "stft_n_frames", "waveform_tile_geometry",
           "ChanDesc", "ImgDesc", "RasterDesc", "WaveDesc", "TileGeom", "LIB_PATH"]
