"""Pins the CPU oracle (oracle/thesia_oracle.c) against every known-answer test the
reference holds for the hot path.  Each test cites the reference
test it restates.  CPU only."""
import math
import struct

import numpy as np
import pytest

from oracle import oracle as orc


# windows.rs:88-91  hann_window_works
def test_hann_periodic_4():
    assert orc.hann(4, False).tolist() == [0.0, 0.5, 1.0, 0.5]


# utils.rs:165-175  pad_works
def test_pad_constant_and_reflect():
    assert orc.pad_constant([1, 2, 3], 1, 2, 10).tolist() == [10, 1, 2, 3, 10, 10]
    assert orc.pad_reflect([1, 2, 3], 3, 4).tolist() == [2, 3, 2, 1, 2, 3, 2, 1, 2, 3]


# stft.rs:173-196  stft_works: impulse(4,@2), win=4 hop=2 n_fft=4 → exact 3x3
def test_stft_impulse_exact():
    x = np.zeros(4, np.float32)
    x[2] = 1.0
    s = orc.perform_stft(x, 4, 2, 4)
    want = np.array([[0, 0, 0], [0.25, -0.25, 0.25], [0.25, -0.25, 0.25]], np.complex64)
    assert s.shape == (3, 3)
    assert np.array_equal(s.astype(np.complex64), want)


# stft.rs:198-203  stft_short_wav: N=2 < win=8 runs; shape only
def test_stft_short_wav_shape():
    x = np.zeros(2, np.float32)
    x[1] = 1.0
    s = orc.perform_stft(x, 8, 6, 8)
    # padded len 2+8=10 → windows(8, stride 6) = 1 frame
    assert s.shape == (1, 5)


# decibel.rs:257-301
def test_dB_scalar_and_array_rules():
    assert abs(orc.dB_from_amp([0.25])[0] - (-12.0412)) < 1e-4
    assert orc.dB_from_amp([0.0])[0] == -np.inf
    assert math.isnan(orc.dB_from_amp([-1.0])[0])
    assert math.isnan(orc.dB_from_amp([np.nan])[0])
    assert abs(orc.dB_from_amp([1.0], ref_value=2.0, amin=0.0)[0] - (-6.0206)) < 1e-4
    a = orc.dB_from_amp([1.0, 0.5, 0.0, -1.0, np.nan], ref_value=1.0, amin=1e-3)
    assert a[0] == 0.0
    assert abs(a[1] + 6.0206) < 1e-4
    assert abs(a[2] + 60.0) < 1e-5
    assert math.isnan(a[3]) and math.isnan(a[4])


# src-common/src/lib.rs:168-174  mel_hz_convert (f64, 1e-14)
def test_mel_hz_convert():
    assert abs(orc.mel_from_hz(100.0) - 1.5) < 1e-14
    assert abs(orc.mel_from_hz(1100.0) - 16.38629404765444) < 1e-14
    assert abs(orc.mel_to_hz(1.0) - 66.66666666666667) < 1e-14
    assert abs(orc.mel_to_hz(16.0) - 1071.1702874944676) < 1e-14


# src-common/src/lib.rs:176-202  mel_works (f64, 1e-8)
def test_mel_fb_filter0():
    sr, n_fft, n_mel = 24000, 2048, 80
    mel0 = [0.0, 0.07852016499598029, 0.15704032999196058, 0.23556049498794085, 0.25,
            0.17147983500401973, 0.09295967000803942, 0.014439505012059144, 0.0]
    fb = orc.calc_mel_fb(sr, n_fft, n_mel, 0.0, None, True, dtype=np.float64)
    assert fb.shape == (n_fft // 2 + 1, n_mel)
    col0 = fb[:, 0]
    want = np.zeros(n_fft // 2 + 1)
    want[: len(mel0)] = mel0
    assert np.max(np.abs(col0 - want)) < 1e-8
    # the f32 instantiation the app uses agrees to f32 precision
    fb32 = orc.calc_mel_fb(sr, n_fft, n_mel, 0.0, None, True, dtype=np.float32)
    assert np.max(np.abs(fb32[:, 0] - want)) < 1e-6


# src-common/src/lib.rs:204-232  mel_default_works (property over 12 sr x n_fft 2^5..2^14)
@pytest.mark.parametrize("sr", [400, 800, 1000, 2000, 4000, 8000, 16000, 24000, 44100, 48000, 88200, 96000])
def test_mel_default_property(sr):
    for e in range(5, 13):  # 2^13, 2^14 covered for two rates below (keeps CPU suite short)
        n_fft = 2 ** e
        n_mel = orc.mel_default_n_mel(sr, n_fft)
        fb = orc.calc_mel_fb(sr, n_fft, n_mel)
        assert np.all(fb.sum(axis=0) > 0), (sr, n_fft, n_mel)
        if n_mel == fb.shape[0]:
            continue
        fb2 = orc.calc_mel_fb(sr, n_fft, n_mel + 1)
        assert np.any(fb2.sum(axis=0) == 0), (sr, n_fft, n_mel)


@pytest.mark.parametrize("sr", [44100, 48000])
def test_mel_default_property_large(sr):
    for n_fft in (8192, 16384):
        n_mel = orc.mel_default_n_mel(sr, n_fft)
        fb = orc.calc_mel_fb(sr, n_fft, n_mel)
        assert np.all(fb.sum(axis=0) > 0)
        fb2 = orc.calc_mel_fb(sr, n_fft, n_mel + 1)
        assert np.any(fb2.sum(axis=0) == 0)


# simd.rs:1111-1138  test_find_min_max_separate (small exact cases incl ±inf, empty, single)
def test_find_min_max_cases():
    assert orc.find_min_max([1, 2, 3, 4, 5]) == (1.0, 5.0)
    assert orc.find_min_max([-1, -2, -3, -4, -5]) == (-5.0, -1.0)
    assert orc.find_min_max([0, 0, 0]) == (0.0, 0.0)
    assert orc.find_min_max([np.inf, -np.inf, 0]) == (-np.inf, np.inf)
    assert orc.find_min_max([1.0]) == (1.0, 1.0)
    assert orc.find_min_max([]) == (np.inf, -np.inf)


# simd.rs:1274-1295  test_sum
def test_sum_cases():
    cases = [([1, 2, 3, 4], 10.0), ([-1, -2, 3], 0.0), ([0, 0, 0], 0.0), ([1.0], 1.0), ([], 0.0),
             ([i - 64.0 for i in range(128)], -64.0)]
    for data, want in cases:
        for mis in range(8):
            assert abs(orc.sum_avx2(data, mis) - want) < 1e-5


# simd.rs:1252-1272  test_sum_squares
def test_sum_squares_cases():
    for data, want in [([1, 2, 3, 4], 30.0), ([-1, -2, -3], 14.0), ([0, 0, 0], 0.0), ([1.0], 1.0), ([], 0.0)]:
        assert abs(orc.sum_squares(data) - want) < 1e-5
    # the compensated sum tracks float64 where a naive f32 sum drifts
    x = np.random.default_rng(0).uniform(-1, 1, 1_000_000).astype(np.float32)
    exact = float(np.sum(x.astype(np.float64) ** 2))
    assert abs(orc.sum_squares(x) - exact) <= 2e-7 * exact


# simd.rs:1357-1379  test_abs_max
def test_abs_max_cases():
    for data, want in [([1, -2, 3, -4], 4.0), ([-1, -2, -3], 3.0), ([0, 0, 0], 0.0), ([1.0], 1.0), ([-1.0], 1.0), ([], 0.0)]:
        assert abs(orc.abs_max(data) - want) < 1e-5


# simd.rs:1438-1456  test_scalar_mul
def test_scalar_mul_cases():
    assert orc.scalar_mul([1, 2, 3, 4], 2.0).tolist() == [2, 4, 6, 8]
    assert orc.scalar_mul([-1, -2, -3], 3.0).tolist() == [-3, -6, -9]
    assert orc.scalar_mul([0, 0, 0], 5.0).tolist() == [0, 0, 0]
    assert orc.scalar_mul([1.0], 0.0).tolist() == [0]
    assert orc.scalar_mul([], 2.0).tolist() == []


# visualize/drawing.rs:41-56  spectrogram_to_img_transposes_and_clamps_dB_values
def test_convert_spectrogram_to_img_known():
    spec = np.array([[-100.0, -50.0, 0.0], [100.0, -200.0, -25.0]], np.float32)
    img = orc.convert_spectrogram_to_img(spec, (0, 4), (-100.0, 0.0), 4)
    assert img.shape == (4, 2)
    assert img.tolist() == [[16384, 65535], [40960, 0], [65535, 53247], [0, 0]]


def test_convert_all_neg_inf_is_zero_image():  # drawing.rs:16-18
    spec = np.full((3, 5), -np.inf, np.float32)
    img = orc.convert_spectrogram_to_img(spec, (0, 5), (-np.inf, -np.inf), 258)
    assert img.shape == (5, 3) and not img.any()


# render_tiles.rs:408-433
def test_waveform_tile_known():
    b = orc.encode_waveform_tile([-1.0, 0.0, 0.5, 1.0], 3, 1, 0)
    assert struct.unpack_from("<Q", b, 0)[0] == 3
    assert struct.unpack_from("<I", b, 8)[0] == 2
    assert struct.unpack_from("<I", b, 12)[0] == 2
    assert struct.unpack_from("<fff", b, 24) == (-1.0, 0.0, -0.5)
    b = orc.encode_waveform_tile(np.full(1025, 0.25, np.float32), 1, 0, 1)
    assert struct.unpack_from("<I", b, 8)[0] == 1
    b = orc.encode_waveform_tile(np.arange(64, dtype=np.float32) - 32.0, 1, 6, 0)
    assert struct.unpack_from("<I", b, 8)[0] == 1
    assert struct.unpack_from("<fff", b, 24) == (-32.0, 31.0, -0.5)


def test_waveform_tile_out_of_range_is_header_only():  # render_tiles.rs:237-241
    b = orc.encode_waveform_tile(np.zeros(10, np.float32), 7, 0, 5)
    assert len(b) == 24 and struct.unpack_from("<I", b, 8)[0] == 0


# render_tiles.rs:435-471
def test_spectrogram_tile_known():
    colors = bytes([0, 0, 0, 255, 255, 0, 0, 255])
    spec = np.array([[0, 65535], [65535, 65535]], np.uint16)
    b = orc.encode_spectrogram_tile(spec, colors, 4, 1, 1, 0, 0)
    assert struct.unpack_from("<II", b, 8) == (1, 1)
    assert b[40:] == bytes([255, 0, 0, 255])

    spec = np.full((513, 513), 65535, np.uint16)
    b = orc.encode_spectrogram_tile(spec, colors, 4, 0, 0, 1, 1)
    assert struct.unpack_from("<II", b, 8) == (5, 5)
    assert struct.unpack_from("<II", b, 32) == (508, 508)
    px = np.frombuffer(b[40:], np.uint8).reshape(-1, 4)
    assert np.all(px == [255, 0, 0, 255])

    spec = np.array([[0], [65535]], np.uint16)
    b = orc.encode_spectrogram_tile(spec, colors, 4, 0, 0, 0, 0)
    assert b[40:44] == bytes([255, 0, 0, 255]) and b[44:48] == bytes([0, 0, 0, 255])


# spectrogram.rs:47-54,56-98 and SURVEY fact 4
def test_framing_params():
    assert orc.calc_framing_params(40.0, 4, 1, 48000) == (480, 1920, 2048)
    assert orc.calc_framing_params(40.0, 4, 1, 44100) == (441, 1764, 2048)
    assert orc.calc_framing_params(2048 / 48, 4, 1, 48000) == (512, 2048, 2048)
    assert orc.calc_framing_params(1024 / 48, 4, 1, 48000) == (256, 1024, 1024)
    assert orc.calc_framing_params(4096 / 48, 4, 1, 48000) == (1024, 4096, 4096)
    assert orc.calc_framing_params(40.0, 4, 2, 48000) == (480, 1920, 4096)


# core/mod.rs:169-180
def test_global_db_range():
    assert orc.global_db_range([-150.0, -30.0], [-3.0, 5.0], 100.0) == (-100.0, 0.0)
    assert orc.global_db_range([-np.inf], [-np.inf], 100.0) == (-np.inf, -np.inf)
    assert orc.global_db_range([-np.inf, -40], [-20.0, -10], 100.0) == (-110.0, -10.0)


# src-common/src/lib.rs:144-159
def test_hz_range_to_idx():
    assert orc.hz_range_to_idx(orc.LINEAR, (0.0, 24000.0), 48000, 1025) == (0, 1025)
    assert orc.hz_range_to_idx(orc.LINEAR, (0.0, 24000.0), 24000, 513) == (0, 1026)
    assert orc.hz_range_to_idx(orc.MEL, (0.0, 24000.0), 48000, 128) == (0, 128)
    assert orc.hz_range_to_idx(orc.LINEAR, (5.0, 5.0), 48000, 128) == (0, 0)
