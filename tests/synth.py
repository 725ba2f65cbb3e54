"""Deterministic synthetic audio: per track a sum of 8 sinusoids (log-uniform
50 Hz..0.45*sr, amplitude 0.02..0.2, random phase) + one linear chirp 100 Hz -> 0.4*sr + uniform
noise +-1e-3, hard-limited to [-1, 1], f32.  Seed = 0x7E51A + track_index."""
import numpy as np

SEED0 = 0x7E51A


def track_params(track_index: int, sr: int):
    rng = np.random.default_rng(SEED0 + track_index)
    freqs = np.exp(rng.uniform(np.log(50.0), np.log(0.45 * sr), 8))
    amps = rng.uniform(0.02, 0.2, 8)
    phases = rng.uniform(0, 2 * np.pi, 8)
    chirp_amp = rng.uniform(0.02, 0.2)
    noise_seed = int(rng.integers(0, 2 ** 31 - 1))
    return freqs, amps, phases, chirp_amp, noise_seed


def synth_track(track_index: int, sr: int, n: int) -> np.ndarray:
    freqs, amps, phases, chirp_amp, noise_seed = track_params(track_index, sr)
    t = np.arange(n, dtype=np.float64) / sr
    x = np.zeros(n, np.float64)
    for f, a, p in zip(freqs, amps, phases):
        x += a * np.sin(2 * np.pi * f * t + p)
    dur = max(n / sr, 1e-9)
    f0, f1 = 100.0, 0.4 * sr
    x += chirp_amp * np.sin(2 * np.pi * (f0 * t + 0.5 * (f1 - f0) / dur * t * t))
    x += np.random.default_rng(noise_seed).uniform(-1e-3, 1e-3, n)
    return np.clip(x, -1.0, 1.0).astype(np.float32)
