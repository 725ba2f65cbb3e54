"""Cross-checks of the oracle against float64 numpy (the mathematical DFT) — the ground truth
for the arithmetic that lives in un-vendored crates (realfft/rustfft, sgemm); SURVEY.md §8c."""
import numpy as np
import pytest

from oracle import oracle as orc


def reflect_index(i, n):
    """periodic extension of numpy 'reflect'."""
    if n == 1:
        return np.zeros_like(i)
    P = 2 * (n - 1)
    j = np.mod(i, P)
    return np.where(j < n, j, P - j)


def simple_frames(x, win, hop):
    """frame k = reflect-padded x[k*hop - win//2 : +win]."""
    n = len(x)
    T = (n + 2 * (win // 2) - win) // hop + 1
    idx = (np.arange(T)[:, None] * hop - win // 2) + np.arange(win)[None, :]
    return x[reflect_index(idx, n)]


def numpy_stft(x, win, hop, n_fft):
    w = orc.calc_normalized_win(win, n_fft)
    fr = (simple_frames(x, win, hop) * w[None, :]).astype(np.float32)  # f32 product like the reference
    pl = (n_fft - win) // 2
    buf = np.zeros((fr.shape[0], n_fft), np.float64)
    buf[:, pl:pl + win] = fr
    return np.fft.rfft(buf, axis=1)


@pytest.mark.parametrize("win,hop,n_fft", [(4, 2, 4), (8, 2, 8), (16, 4, 16), (64, 16, 64), (30, 10, 32),
                                           (1920, 480, 2048), (2048, 512, 2048), (15, 5, 16), (9, 3, 16),
                                           (1024, 256, 1024), (4096, 1024, 4096), (8, 8, 8), (6, 1, 8)])
def test_three_segment_framing_equals_simple_formula(win, hop, n_fft):
    rng = np.random.default_rng(win * 1000 + hop)
    for n in sorted({win, win + 1, win + hop - 1, win + hop, 2 * win + 3, 5 * win + hop // 2 + 1, 3 * win - 1,
                     max(2, win // 2), max(2, win // 2 + 1), max(2, win - 1), 2, 3, 7 * hop}):
        x = rng.standard_normal(n).astype(np.float32)
        got = orc.perform_stft(x, win, hop, n_fft)
        want = numpy_stft(x, win, hop, n_fft)
        assert got.shape == want.shape, (n, got.shape, want.shape)
        scale = np.abs(want).max() + 1e-30
        assert np.abs(got - want).max() <= 2e-6 * scale, (n, win, hop)


@pytest.mark.parametrize("win,hop,n_fft", [(4, 1, 12), (16, 4, 48), (30, 10, 96), (64, 16, 192), (128, 32, 640), (1920, 480, 6144),
                                           (1920, 480, 10240), (100, 25, 128 * 7), (512, 128, 512 * 63)])
def test_oracle_dft_of_lengths_that_are_not_powers_of_two(win, hop, n_fft):
    """f_overlap = 3, 5, 6, 7, 63 (spectrogram.rs:66-72): the oracle's DFT of n_fft = 2^a * odd — direct O(N^2) sum below 64
    points, interleaved sub-transforms above — is the f64 DFT (numpy.fft.rfft)."""
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(2 * n_fft + 3 * hop + 1).astype(np.float32)
    got = orc.perform_stft(x, win, hop, n_fft)
    want = numpy_stft(x, win, hop, n_fft)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 2e-6 * np.abs(want).max()


def test_frame_count_formula_sweep():
    for win in range(2, 40):
        for hop in range(1, win + 1):
            for n in range(2, 3 * win + 5):
                T = orc.stft_n_frames(n, win, hop)
                assert T == (n + 2 * (win // 2) - win) // hop + 1, (n, win, hop)


@pytest.mark.parametrize("n_fft", [4, 8, 64, 1024, 2048, 4096])
def test_f32_fft_matches_f64(n_fft):
    rng = np.random.default_rng(n_fft)
    x = rng.standard_normal(n_fft * 3 + 17).astype(np.float32)
    a = orc.perform_stft(x, n_fft, n_fft // 4, n_fft, fft32=False)
    b = orc.perform_stft(x, n_fft, n_fft // 4, n_fft, fft32=True)
    assert np.abs(a - b).max() <= 3e-6 * np.abs(a).max()


def test_calc_spec_linear_and_mel_vs_float64():
    rng = np.random.default_rng(7)
    sr, win, hop, n_fft = 48000, 1920, 480, 2048
    t = np.arange(20000) / sr
    x = (0.3 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(t.size)).astype(np.float32)
    spec, amp = orc.calc_spec(x, win, hop, n_fft, return_amp=True)
    want_amp = np.abs(numpy_stft(x, win, hop, n_fft))
    assert np.abs(amp - want_amp).max() <= 1e-6 * want_amp.max()
    want_db = 20 * np.log10(want_amp)
    ok = want_amp >= 1e-5 * want_amp.max()
    assert np.abs(spec - want_db)[ok].max() < 1e-3
    fb = orc.calc_mel_fb(sr, n_fft, 128)
    mspec = orc.calc_spec(x, win, hop, n_fft, mel_fb=fb)
    want_mel = want_amp @ orc.calc_mel_fb(sr, n_fft, 128, dtype=np.float64)
    assert np.abs(mspec - 20 * np.log10(want_mel)).max() < 1e-3


def test_silence_gives_neg_inf():
    spec = orc.calc_spec(np.zeros(5000, np.float32), 1024, 256, 1024)
    assert np.all(np.isneginf(spec))


def test_oracle_matches_committed_f64_fixtures(golden_dir):
    z = np.load(f"{golden_dir}/stft_f64_cases.npz")
    names = sorted({k.rsplit("_", 1)[0] for k in z.files})
    assert len(names) == 7
    for name in names:
        sr, win, hop, n_fft = (int(v) for v in z[name + "_par"])
        _, amp = orc.calc_spec(z[name + "_x"], win, hop, n_fft, return_amp=True)
        want = z[name + "_amp"]
        assert amp.shape == want.shape
        assert np.abs(amp - want).max() <= 1e-6 * want.max()
