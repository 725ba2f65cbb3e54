"""CPU lane emulation of the wave STFT kernel (tests/emu, built from the same stft_wave.h the
gfx950 kernel uses) against the oracle: checks the Stockham index arithmetic, the LDS swizzle,
the mirror exchange and the split pass without a GPU.  CPU only."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as orc
from tests.synth import synth_track

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def emu():
    subprocess.check_call(["make", "-C", os.path.join(HERE, "emu"), "-s"])
    lib = C.CDLL(os.path.join(HERE, "emu", "_build", "libemu_stft.so"))
    f32p = C.POINTER(C.c_float)
    lib.emu_stft_wave.argtypes = [f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, C.c_uint32, f32p]
    return lib


@pytest.mark.parametrize("win,hop,n_fft,n", [(2048, 512, 2048, 9000), (1920, 480, 2048, 9000), (1764, 441, 2048, 7000),
                                             (2048, 512, 2048, 700), (1024, 256, 1024, 5000), (1000, 250, 1024, 5000),
                                             (4096, 1024, 4096, 14000), (3001, 3001, 4096, 14000),
                                             (2047, 2047, 2048, 9000)])
def test_emulated_wave_kernel_matches_oracle(emu, win, hop, n_fft, n):
    x = synth_track(n_fft + win, 48000, n)
    w = orc.calc_normalized_win(win, n_fft)
    T = orc.stft_n_frames(n, win, hop)
    out = np.empty((T, n_fft // 2 + 1), np.float32)
    f32p = C.POINTER(C.c_float)
    rc = emu.emu_stft_wave(x.ctypes.data_as(f32p), n, win, hop, n_fft, w.ctypes.data_as(f32p), T,
                           out.ctypes.data_as(f32p))
    assert rc == 0
    want, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
    interior = ~np.isnan(out[:, 0])  # the wave kernel takes interior frames only (edges -> generic kernel)
    e0 = np.arange(T) * hop - win // 2 - (n_fft - win) // 2     # start of the frame's n_fft-sample span
    assert np.array_equal(interior, (e0 >= 0) & (e0 + n_fft <= n))
    if n >= 2 * win:
        assert interior.sum() >= T - 6
    got_amp = np.power(10.0, out[interior].astype(np.float64) / 20.0)
    rel = (np.abs(got_amp - amp[interior]) / amp[interior].max(axis=1, keepdims=True)).max() if interior.any() else 0
    assert rel <= 2e-6, rel


def test_fma_corrected_division_is_exact(emu):
    """The quantiser divides by the launch-uniform span with one FMA correction step on the correctly rounded
    reciprocal (kernels_image.hip: quantise).  Sampled version of the exhaustive check (all 2^32 dividends x 60
    divisors: zero mismatches) that was run when the kernel was written."""
    lib = emu
    lib.emu_fma_div_mismatches.restype = C.c_uint64
    lib.emu_fma_div_mismatches.argtypes = [C.POINTER(C.c_float), C.c_uint64, C.c_float]
    rng = np.random.default_rng(5)
    n = 1 << 21
    # dividends as the kernel sees them: dB - min_dB, plus random bit patterns of moderate magnitude
    a = np.concatenate([rng.uniform(-300, 300, n).astype(np.float32),
                        (rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)).view(np.float32)])
    a = a[np.isfinite(a) & ((a == 0) | ((np.abs(a) > 1e-30) & (np.abs(a) < 1e30)))]
    a = np.ascontiguousarray(a)
    for b in [100.0, 99.99999, 60.0, 3.0, 1e-3, 2e3] + list(np.exp(rng.uniform(np.log(1e-3), np.log(2e3), 10))):
        bad = lib.emu_fma_div_mismatches(a.ctypes.data_as(C.POINTER(C.c_float)), a.size, np.float32(b))
        assert bad == 0, (b, bad)
