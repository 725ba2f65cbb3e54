"""Host-side logic of the product library (plan preparation tables, shapes, tile geometry)
against the oracle.  These entry points need no GPU.  CPU only."""
import struct

import numpy as np
import pytest

import thesia_amd as ta
from oracle import oracle as orc


@pytest.mark.parametrize("sr", [8000, 16000, 22050, 24000, 44100, 48000, 96000])
def test_framing_params_grid(sr):
    for win_ms in (1.0, 5.0, 10.0, 2048 / 48, 40.0, 46.4, 64.0, 100.0):
        for t_ov in (1, 2, 4, 8, 16, 32):
            for f_ov in (1, 2):
                assert ta.calc_framing_params(win_ms, t_ov, f_ov, sr) == orc.calc_framing_params(win_ms, t_ov, f_ov, sr)


def test_framing_params_rejects_bad_setting():
    for bad in ((0.0, 4, 1), (40.0, 0, 1), (40.0, 4, 0)):
        with pytest.raises(ta.ThError):
            ta.calc_framing_params(*bad, 48000)


def test_n_frames_sweep():
    for win in range(2, 33):
        for hop in range(1, win + 1):
            for n in range(1, 3 * win + 3):
                if n >= 2:
                    assert ta.stft_n_frames(n, win, hop) == orc.stft_n_frames(n, win, hop), (n, win, hop)


@pytest.mark.parametrize("win,n_fft", [(4, 4), (1920, 2048), (2048, 2048), (1764, 2048), (1024, 1024), (4096, 4096),
                                       (15, 16), (480, 512)])
def test_window_bit_exact(win, n_fft):
    assert np.array_equal(ta.calc_normalized_win(win, n_fft), orc.calc_normalized_win(win, n_fft))


@pytest.mark.parametrize("sr,n_fft,n_mel", [(24000, 2048, 80), (44100, 2048, 128), (48000, 2048, 128), (48000, 1024, 40),
                                            (16000, 512, 64), (48000, 4096, 256)])
def test_mel_fb_bit_exact(sr, n_fft, n_mel):
    a, b = ta.calc_mel_fb(sr, n_fft, n_mel), orc.calc_mel_fb(sr, n_fft, n_mel)
    assert np.array_equal(a, b)
    assert np.count_nonzero(a, axis=1).max() <= 2  # each frequency row feeds at most 2 filters


@pytest.mark.parametrize("sr", [8000, 16000, 22050, 44100, 48000, 96000])
def test_mel_default_count(sr):
    for n_fft in (256, 1024, 2048, 4096):
        assert ta.mel_default_n_mel(sr, n_fft) == orc.mel_default_n_mel(sr, n_fft)


def test_hz_range_to_idx():
    for scale in (ta.LINEAR, ta.MEL):
        for sr, n in ((48000, 1025), (44100, 1025), (8000, 257), (48000, 347), (22050, 128)):
            for max_sr in (48000, 96000, 44100, 8000):
                rng = (0.0, max_sr / 2.0)
                assert ta.hz_range_to_idx(scale, rng, sr, n) == orc.hz_range_to_idx(scale, rng, sr, n)
    assert ta.hz_range_to_idx(ta.LINEAR, (3.0, 3.0), 48000, 10) == (0, 0)


def test_global_db_range():
    for mins, maxs, r in (([-150, -30], [-3, 5], 100), ([-np.inf], [-np.inf], 100), ([-np.inf, -40], [-20, -10], 80),
                          ([], [], 100)):
        assert ta.global_db_range(mins, maxs, r) == orc.global_db_range(mins, maxs, r)


def test_spectrogram_tile_geometry_matches_oracle_headers():
    cm = bytes([0, 0, 0, 255, 255, 255, 255, 255])
    for (hh, w) in ((2, 2), (513, 513), (1025, 5626), (1, 1), (600, 40), (1025, 2813)):
        img = np.zeros((hh, w), np.uint16)
        for lx, ly, tx, ty in ((0, 0, 0, 0), (0, 0, 1, 1), (1, 1, 0, 0), (2, 1, 1, 0), (0, 0, 10, 2), (0, 0, 99, 99),
                               (3, 3, 0, 0), (40, 70, 0, 0)):
            g = ta.spectrogram_tile_geometry(w, hh, lx, ly, tx, ty)
            if (lx or ly) and g.width * g.height > 0 and (lx < 32 and ly < 32):
                pass
            b = orc.encode_spectrogram_tile(img, cm, 1, lx, ly, tx, ty) if (lx < 8 and ly < 8) else None
            if b is not None:
                assert struct.unpack_from("<II", b, 8) == (g.width, g.height)
                assert struct.unpack_from("<II", b, 32) == (g.origin_x, g.origin_y)


def test_waveform_tile_geometry_matches_oracle_headers():
    for n in (1, 4, 1024, 1025, 48000, 2113529):
        wav = np.zeros(n, np.float32)
        for level in (0, 1, 5, 6, 10, 20, 40, 70):
            for tile in (0, 1, 2, 2064, 2 ** 31):
                start, bins, spb = ta.waveform_tile_geometry(n, level, tile)
                b = orc.encode_waveform_tile(wav, 9, level, tile)
                assert struct.unpack_from("<I", b, 8)[0] == bins
                assert struct.unpack_from("<I", b, 12)[0] == min(spb, 2 ** 32 - 1)
                assert len(b) == 24 + 12 * bins
