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
    lib.emu_stft_wave_multi.argtypes = lib.emu_stft_wave.argtypes
    lib.emu_stft_wave_pk.argtypes = lib.emu_stft_wave.argtypes
    return lib


@pytest.mark.parametrize("win,hop,n_fft,n", [(2048, 512, 2048, 9000), (1920, 480, 2048, 9000), (1764, 441, 2048, 7000),
                                             (2048, 512, 2048, 700), (1024, 256, 1024, 5000), (1000, 250, 1024, 5000),
                                             (4096, 1024, 4096, 14000), (3001, 3001, 4096, 14000),
                                             (2047, 2047, 2048, 9000),
                                             # the workgroup-per-frame plan (stft_block.h)
                                             (8192, 2048, 8192, 30000), (7680, 1920, 8192, 30000), (16384, 4096, 16384, 60000),
                                             (15001, 5000, 16384, 70000), (32768, 8192, 32768, 100000), (19200, 4800, 32768, 80000),
                                             # n_fft 65536: the same plan with planar exchanges (round 5)
                                             (65536, 16384, 65536, 200000), (48000, 12000, 65536, 160000)])
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
        assert interior.sum() >= T - 8  # (win < n_fft at hop = win / 4: up to four boundary frames on either side)
    got_amp = np.power(10.0, out[interior].astype(np.float64) / 20.0)
    rel = (np.abs(got_amp - amp[interior]) / amp[interior].max(axis=1, keepdims=True)).max() if interior.any() else 0
    assert rel <= 2e-6, rel


@pytest.mark.parametrize("win,hop,n", [(2048, 512, 9000), (1920, 480, 9000), (1764, 441, 7000), (2047, 2047, 9000)])
def test_emulated_packed_pipeline_matches_oracle_and_scalar_plan(emu, win, hop, n):
    """stft_pk.h / WaveFft<10>::*_pk (round 4): the n_fft 2048 plan on register pairs (v_pk_*_f32 on the GPU).  The lane
    functions run on the CPU with the pair helpers as plain structs: every bin emitted exactly once, the oracle's
    magnitudes, and the scalar plan's to rounding (the butterfly algebra is the same; only d = a - t b of the cross levels
    is formed directly instead of as 2 a - s)."""
    n_fft = 2048
    x = synth_track(n_fft + win + 3, 48000, n)
    w = orc.calc_normalized_win(win, n_fft)
    T = orc.stft_n_frames(n, win, hop)
    f32p = C.POINTER(C.c_float)
    out, ref = np.empty((T, n_fft // 2 + 1), np.float32), np.empty((T, n_fft // 2 + 1), np.float32)
    assert emu.emu_stft_wave_pk(x.ctypes.data_as(f32p), n, win, hop, n_fft, w.ctypes.data_as(f32p), T, out.ctypes.data_as(f32p)) == 0
    assert emu.emu_stft_wave(x.ctypes.data_as(f32p), n, win, hop, n_fft, w.ctypes.data_as(f32p), T, ref.ctypes.data_as(f32p)) == 0
    want, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
    interior = ~np.isnan(ref[:, 0])
    assert interior.any() and np.array_equal(interior, ~np.isnan(out[:, 0]))
    assert not np.isnan(out[interior]).any()   # every bin of every interior frame emitted exactly once
    got_amp = np.power(10.0, out[interior].astype(np.float64) / 20.0)
    ref_amp = np.power(10.0, ref[interior].astype(np.float64) / 20.0)
    scale = amp[interior].max(axis=1, keepdims=True)
    assert (np.abs(got_amp - amp[interior]) / scale).max() <= 2e-6
    assert (np.abs(got_amp - ref_amp) / scale).max() <= 3e-6  # (two f32 FFTs with different roundings: each within 2e-6 of the f64 truth)


@pytest.mark.parametrize("win,hop,n_fft,n", [(1024, 256, 1024, 5000), (1000, 250, 1024, 5003), (512, 128, 512, 3000),
                                             (320, 80, 512, 2500), (480, 120, 512, 2600), (511, 300, 512, 4000)])
def test_emulated_multi_frame_plan_matches_oracle(emu, win, hop, n_fft, n):
    """stft_wave_multi.h: two (n_fft 1024) or four (n_fft 512) frames per wave, 16 points per lane, plane exchanges —
    the lane functions run on the CPU against the oracle."""
    x = synth_track(n_fft + win + 1, 48000, n)
    w = orc.calc_normalized_win(win, n_fft)
    T = orc.stft_n_frames(n, win, hop)
    out = np.empty((T, n_fft // 2 + 1), np.float32)
    f32p = C.POINTER(C.c_float)
    rc = emu.emu_stft_wave_multi(x.ctypes.data_as(f32p), n, win, hop, n_fft, w.ctypes.data_as(f32p), T, out.ctypes.data_as(f32p))
    assert rc == 0
    want, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
    interior = ~np.isnan(out[:, 0])
    e0 = np.arange(T) * hop - win // 2 - (n_fft - win) // 2
    assert np.array_equal(interior, (e0 >= 0) & (e0 + n_fft <= n)) and interior.sum() >= T - 8
    assert not np.isnan(out[interior]).any()   # every bin of every interior frame emitted exactly once
    got_amp = np.power(10.0, out[interior].astype(np.float64) / 20.0)
    rel = (np.abs(got_amp - amp[interior]) / amp[interior].max(axis=1, keepdims=True)).max()
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


@pytest.mark.parametrize("sr,n_fft,n_mel", [(44100, 2048, 128), (48000, 2048, 0), (44100, 2048, 0), (48000, 1024, 128),
                                            (24000, 2048, 80), (48000, 2048, 40), (16000, 1024, 0), (96000, 2048, 256),
                                            (48000, 2048, 400), (44100, 2048, 17)])
def test_fused_mel_epilogue_tables(emu, sr, n_fft, n_mel):
    """mel_fuse.h piece / gather tables + the lane functions of the fused mel epilogue against amp @ fb in f64
    (spectrogram.rs:207).  n_mel = 0: the reference's default count (src-common/src/lib.rs:91-103)."""
    fb = orc.calc_mel_fb_default(sr, n_fft) if n_mel == 0 else orc.calc_mel_fb(sr, n_fft, n_mel)
    fb = np.ascontiguousarray(fb, np.float32)
    F, M = fb.shape
    assert F == n_fft // 2 + 1
    rng = np.random.default_rng(3)
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    emu.emu_mel_fuse.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p]
    info = np.zeros(3, np.uint32)
    for trial in range(3):
        # wide dynamic range, like a spectrum: quiet bands next to loud ones
        amp = (rng.uniform(0, 1, F) * 10.0 ** rng.uniform(-6, 0, F)).astype(np.float32)
        out = np.empty(M, np.float32)
        rc = emu.emu_mel_fuse(amp.ctypes.data_as(f32p), fb.ctypes.data_as(f32p), F, M, 512, out.ctypes.data_as(f32p),
                              info.ctypes.data_as(u32p))
        assert rc == 0, "filterbank not fusable"  # (needs <= 512 pieces and <= 512 mels; else the plan falls back)
        want = amp.astype(np.float64) @ fb.astype(np.float64)
        # every term is non-negative: the sum is well conditioned, f32 accumulation order is the only difference
        assert np.all(np.abs(out - want) <= 4e-6 * want + 1e-30), np.abs(out / np.maximum(want, 1e-300) - 1).max()
    # the pieces cost about what the non-zeros do: 2 weights per bin, 4 bins per piece
    assert info[0] <= F / 4 + M + 2, info


@pytest.mark.parametrize("sr,n_fft,n_mel", [(44100, 2048, 128), (48000, 2048, 0), (44100, 2048, 0), (16000, 1024, 0),
                                            (22050, 1024, 0), (48000, 1024, 80), (32000, 2048, 0), (48000, 2048, 512)])
def test_banded_mel_epilogue_matches_the_dense_product(emu, sr, n_fft, n_mel):
    """build_mel_band + mel_banded (lane = mel, taps to the widest filter of each group of 64 mels) against amp @ fb in f64, in the
    table's three layouts: first bins as the filters start, spread over the LDS banks, paired (8-byte amplitude reads)."""
    M = n_mel or orc.mel_default_n_mel(sr, n_fft)
    fb = np.ascontiguousarray(orc.calc_mel_fb(sr, n_fft, M), np.float32)
    F = n_fft // 2 + 1
    rng = np.random.default_rng(5)
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    emu.emu_mel_band.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p]
    infos = []
    for layout in (0, 1, 2):
        info = np.zeros(5, np.uint32)
        for trial in range(3):
            amp = (rng.uniform(0, 1, F) * 10.0 ** rng.uniform(-6, 0, F)).astype(np.float32)
            out = np.empty(M, np.float32)
            rc = emu.emu_mel_band(amp.ctypes.data_as(f32p), fb.ctypes.data_as(f32p), F, M, 1 << 16, layout, out.ctypes.data_as(f32p),
                                  info.ctypes.data_as(u32p))
            assert rc == 0, (layout, rc)
            want = amp.astype(np.float64) @ fb.astype(np.float64)
            assert np.all(np.abs(out - want) <= 4e-6 * want + 1e-30), np.abs(out / np.maximum(want, 1e-300) - 1).max()
        assert info[1] == (M + 63) // 64 and info[2] % 4 == 0
        infos.append(info)
    # the bank spreading never costs LDS cycles, and the paired layout (half the cycles per conflict-free tap, a few more taps)
    # stays below the plain one as well; neither adds more than 8 taps to a group
    assert infos[1][4] <= infos[0][4] and infos[1][3] <= infos[0][3] + 8 * infos[0][1]
    assert infos[2][4] <= infos[0][4] and infos[2][3] <= infos[0][3] + 8 * infos[0][1], (infos[0], infos[2])
    # a filterbank with a filter wider than 128 bins has no table (the plan keeps the pieces or the matrix cores)
    wide = np.zeros((F, 4), np.float32)
    wide[: F // 2, 0] = 1.0
    wide[F // 2:, 1] = 1.0
    assert emu.emu_mel_band(amp.ctypes.data_as(f32p), wide.ctypes.data_as(f32p), F, 4, 1 << 16, 1, out.ctypes.data_as(f32p),
                            info.ctypes.data_as(u32p)) == 1


@pytest.mark.parametrize("sr,n_fft,n_mel", [(48000, 2048, 0), (44100, 2048, 128), (44100, 2048, 0), (32000, 2048, 0), (48000, 2048, 100),
                                            (48000, 2048, 321)])
def test_banded_mel_frame_pairs_match_the_one_frame_sums(emu, sr, n_fft, n_mel):
    """mel_banded_pair (round 5: two amplitude rows, one pass over the paired table) == mel_banded on each row, bit for bit, and the
    extremes over ALL 64 lanes of every group equal the extremes over the real mels (lanes past the last mel repeat the last
    filter, so the kernel masks only their stores)."""
    M = n_mel or orc.mel_default_n_mel(sr, n_fft)
    fb = np.ascontiguousarray(orc.calc_mel_fb(sr, n_fft, M), np.float32)
    F = n_fft // 2 + 1
    rng = np.random.default_rng(11)
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    emu.emu_mel_band.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p]
    emu.emu_mel_band_pair.argtypes = [f32p, f32p, f32p, C.c_uint32, C.c_uint32, f32p, f32p, f32p, u32p]
    for trial in range(3):
        amps = [(rng.uniform(0, 1, F) * 10.0 ** rng.uniform(-6, 0, F)).astype(np.float32) for _ in range(2)]
        pair = [np.empty(M, np.float32) for _ in range(2)]
        mm, reach = np.zeros(4, np.float32), np.zeros(1, np.uint32)
        rc = emu.emu_mel_band_pair(amps[0].ctypes.data_as(f32p), amps[1].ctypes.data_as(f32p), fb.ctypes.data_as(f32p), F, M,
                                   pair[0].ctypes.data_as(f32p), pair[1].ctypes.data_as(f32p), mm.ctypes.data_as(f32p), reach.ctypes.data_as(u32p))
        assert rc == 0, (rc, reach)
        assert F <= reach[0] + 64 and reach[0] <= F + 59
        for i in range(2):
            one, info = np.empty(M, np.float32), np.zeros(5, np.uint32)
            assert emu.emu_mel_band(amps[i].ctypes.data_as(f32p), fb.ctypes.data_as(f32p), F, M, 1 << 16, 2, one.ctypes.data_as(f32p),
                                    info.ctypes.data_as(u32p)) == 0
            assert np.array_equal(one, pair[i])
            assert mm[2 * i] == one.min() and mm[2 * i + 1] == one.max()


def test_fused_mel_tables_refuse_what_is_not_a_triangle_filterbank(emu):
    """build_mel_fuse (mel_fuse.h) verifies the structure it relies on and reports "not fusable" (the plan then keeps the
    matrix-core path) for: a dense matrix, three filters at one bin, two non-neighbouring filters at one bin, segments
    that go backwards, a negative weight, too many pieces for the (r, f) buffer."""
    f32p, u32p = C.POINTER(C.c_float), C.POINTER(C.c_uint32)
    emu.emu_mel_fuse.argtypes = [f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p]
    F, M = 257, 20
    good = np.ascontiguousarray(orc.calc_mel_fb(16000, 512, M), np.float32)
    assert good.shape == (F, M)

    def rc(fb, max_pieces=512):
        fb = np.ascontiguousarray(fb, np.float32)
        amp, out, info = np.ones(F, np.float32), np.empty(M, np.float32), np.zeros(3, np.uint32)
        return emu.emu_mel_fuse(amp.ctypes.data_as(f32p), fb.ctypes.data_as(f32p), F, M, max_pieces,
                                out.ctypes.data_as(f32p), info.ctypes.data_as(u32p))

    assert rc(good) == 0
    assert rc(np.random.default_rng(0).uniform(0.1, 1, (F, M))) == 1            # dense
    bad = good.copy(); k = int(np.argmax(good[:, 10])); bad[k, 3] = 0.5; bad[k, 4] = 0.5
    assert rc(bad) == 1                                                         # more filters at one bin
    bad = np.zeros_like(good); bad[50, 2] = 1; bad[50, 7] = 1
    assert rc(bad) == 1                                                         # two that are not neighbours
    bad = np.zeros_like(good); bad[10, 9] = 1; bad[20, 3] = 1
    assert rc(bad) == 1                                                         # segments go backwards
    bad = good.copy(); bad[k, 10] = -bad[k, 10]
    assert rc(bad) == 1                                                         # negative weight
    assert rc(good, max_pieces=8) == 1                                          # does not fit the (r, f) buffer


MOMENT_CASES = [(96000, 4096, 0), (88200, 4096, 0), (48000, 4096, 0), (48000, 4096, 128), (44100, 4096, 64), (48000, 2048, 0),
                (16000, 1024, 0), (8000, 512, 0), (48000, 8192, 0), (48000, 4096, 1000), (22050, 4096, 512), (48000, 4096, 5),
                (192000, 4096, 0), (8000, 4096, 0), (32000, 4096, 0), (48000, 16384, 0), (48000, 16384, 40), (96000, 8192, 200)]


def _mom_args(emu):
    f32p, u32p, f64p = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    emu.emu_mel_moments.argtypes = [f32p, C.c_uint32, f32p, f32p, f32p, C.c_uint32, C.c_uint32, C.c_uint32, f32p, u32p, f64p]
    return f32p, u32p, f64p


@pytest.mark.parametrize("sr,n_fft,n_mel", MOMENT_CASES)
def test_moment_form_of_the_mel_filterbank(emu, sr, n_fft, n_mel):
    """build_mel_moments + mel_mom_lane / mel_mom_w_lane / mel_mom_combine (round 6: lane = segment of the triangle points; wide
    segments as moments — per bin two additions, no weight table — narrow ones as their one or two weight pairs) against amp @ fb
    in f64.  In the moment groups the reference's f32 weights are replaced by the line through the f32 triangle points (the table
    additionally carries the rounding of the f32 bin frequencies): `max_dev` is that difference in units of an unnormalised
    weight and `max_amp` how far a filter's 1 / d enlarges the rounding of the two terms the line is evaluated from; a result may
    differ from the table's product by (max_dev + a few ulp * max_amp) * (amplitude under the filter) / d plus f32 summation
    error — asserted mel by mel, also for isolated spectral lines (one term per mel: nothing averages out).  NaN behind the
    amplitude row must not reach any sum.  The emulator also walks the workgroup-per-frame kernels' lane table
    (build_mel_mom_lanes; windows instead of masks, batches in lockstep) and wants the same bits."""
    M = n_mel or orc.mel_default_n_mel(sr, n_fft)
    fb = np.ascontiguousarray(orc.calc_mel_fb(sr, n_fft, M), np.float32)
    lin, mf = orc.mel_fb_points(sr, n_fft, M)
    F = n_fft // 2 + 1
    slab_len = 2 * (F - 1 + (F - 1) // 16)   # floats of WaveFft::SLAB_LEN complex slots
    rng = np.random.default_rng(17)
    f32p, u32p, f64p = _mom_args(emu)
    fb64 = fb.astype(np.float64)
    # 1 / d_m (lib.rs:84-86) from the points themselves, f64
    l64, p64 = lin.astype(np.float64), mf.astype(np.float64)
    inv_d, under = np.zeros(M), np.zeros((F, M))
    for m in range(M):
        up = (l64 > p64[m]) & (l64 <= p64[m + 1])
        dn = (l64 > p64[m + 1]) & (l64 <= p64[m + 2])   # (a bin ON the upper point has weight 0 but sits in the segment's sums)
        d = ((l64[up] - p64[m]) / (p64[m + 1] - p64[m])).sum() + ((p64[m + 2] - l64[dn]) / (p64[m + 2] - p64[m + 1])).sum()
        inv_d[m] = 1.0 / d if d > 0 else 0.0
        under[up | dn, m] = 1.0
    infos = []
    for spread in (0, 1):
        info, dev = np.zeros(6, np.uint32), np.zeros(2, np.float64)
        for trial in range(5):
            amp = (rng.uniform(0, 1, F) * 10.0 ** rng.uniform(-6, 0, F)).astype(np.float32)
            if trial >= 3:
                amp[:] = 0.0
                amp[rng.integers(1, F, 9 if trial == 3 else 200)] = 1.0   # isolated lines
            slab = np.full(slab_len, np.nan, np.float32)
            slab[:F] = amp
            out = np.empty(M, np.float32)
            rc = emu.emu_mel_moments(slab.ctypes.data_as(f32p), slab_len, fb.ctypes.data_as(f32p), lin.ctypes.data_as(f32p),
                                     mf.ctypes.data_as(f32p), F, M, spread, out.ctypes.data_as(f32p), info.ctypes.data_as(u32p),
                                     dev.ctypes.data_as(f64p))
            assert rc == 0, rc
            assert not np.isnan(out).any()
            want = amp.astype(np.float64) @ fb64
            mass = amp.astype(np.float64) @ under    # amplitude in the filter's two segments
            bound = (1.5 * dev[0] + 4e-7 * (1.0 + dev[1])) * mass * inv_d + 4e-6 * want + 1e-30
            assert np.all(np.abs(out - want) <= bound), (np.abs(out - want) / np.maximum(bound, 1e-300)).max()
            # the north-star scale (1e-4 of the frame's largest value): an order inside even for lone lines at sample rates whose
            # bin spacing is not an f32 number (44.1 kHz family: max_dev ~ 1e-5; 48 kHz family: ~ 1e-7, the f32 sums of wide filters then dominate)
            assert np.abs(out - want).max() <= (1e-5 if dev[0] > 5e-7 else 2e-6) * want.max() + 1e-30, np.abs(out - want).max() / want.max()
        assert dev[0] < 2e-5 and dev[1] <= 8.0, dev
        assert info[1] == (M + 1 + 63) // 64 and info[4] <= slab_len
        infos.append(info.copy())
    assert infos[1][2] <= infos[0][2] + 4 * infos[0][1]   # spreading adds at most two taps (one unroll step) per group
    assert infos[1][5] == infos[0][5]


def test_moment_form_refuses_what_is_not_a_triangle_filterbank(emu):
    sr, n_fft, M = 48000, 2048, 64
    fb = np.ascontiguousarray(orc.calc_mel_fb(sr, n_fft, M), np.float32)
    lin, mf = orc.mel_fb_points(sr, n_fft, M)
    fb[np.nonzero(fb[:, 10])[0][0], 10] *= 1.5   # not the reference's weight any more
    F = n_fft // 2 + 1
    f32p, u32p, f64p = _mom_args(emu)
    slab = np.zeros(4 * F, np.float32)
    out, info, dev = np.empty(M, np.float32), np.zeros(6, np.uint32), np.zeros(2, np.float64)
    assert emu.emu_mel_moments(slab.ctypes.data_as(f32p), slab.size, fb.ctypes.data_as(f32p), lin.ctypes.data_as(f32p), mf.ctypes.data_as(f32p),
                               F, M, 1, out.ctypes.data_as(f32p), info.ctypes.data_as(u32p), dev.ctypes.data_as(f64p)) == 1


@pytest.mark.parametrize("sr,win_ms,t_overlap,f_overlap", [(4000, 0.5, 1, 67), (8000, 2.0, 4, 65), (16000, 8.0, 2, 71), (8000, 2.0, 2, 130),
                                                           (4000, 0.5, 1, 3), (48000, 5.0, 4, 1)])
def test_chirp_z_plan_on_the_cpu(emu, sr, win_ms, t_overlap, f_overlap):
    """stft_bluestein_kernel's phases (stft_core.h: bluestein_load / _pass / _product / _unchirp / _split) over the tables the plan
    uploads (host_math.cpp: bluestein_tables) — the path n_fft takes when its odd factor is above 63 (f_overlap 65, 67, 71, 130;
    spectrogram.rs:66-72) — against the oracle's f64 DFT, boundary frames (reflect padding) and a channel shorter than the window
    included.  The algorithm does not care what n_fft is: a small odd factor and a power of two go through it here as well.
    Double precision throughout: the amplitudes are the oracle's to f32 rounding."""
    hop, win, n_fft = orc.calc_framing_params(win_ms, t_overlap, f_overlap, sr)
    emu.emu_stft_bluestein.argtypes = emu.emu_stft_wave.argtypes
    f32p = C.POINTER(C.c_float)
    w = orc.calc_normalized_win(win, n_fft)
    for n in (3 * n_fft + 17, max(2, win // 3)):
        x = synth_track(300 + n_fft, sr, n)
        T = orc.stft_n_frames(n, win, hop)
        out = np.empty((T, n_fft // 2 + 1), np.float32)
        M = emu.emu_stft_bluestein(x.ctypes.data_as(f32p), n, win, hop, n_fft, w.ctypes.data_as(f32p), T, out.ctypes.data_as(f32p))
        assert M >= n_fft - 1 and M & (M - 1) == 0 and M < 2 * n_fft
        _, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
        assert amp.shape == out.shape
        assert np.abs(out - amp).max() <= 2e-7 * amp.max() + 1e-30, np.abs(out - amp).max() / amp.max()

