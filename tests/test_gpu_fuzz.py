"""The randomised parity checks of tests/fuzzers.py inside `pytest -m gpu`: fixed seeds, about 200 cases each, a time cap
per fuzzer (the fuzz evidence has to be something the driver runs)."""
import pytest

import thesia_amd as ta
from tests import fuzzers

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = ta.Context(0)
    yield c
    c.close()


def test_fuzz_stft_wave_vs_generic(ctx):
    r = fuzzers.fuzz_stft(ctx, seed=20261003, max_cases=200, max_seconds=90.0)
    assert r["cases"] >= 50 and r["worst"] <= 5e-6, r


def test_fuzz_stft_long_channels(ctx):
    r = fuzzers.fuzz_stft(ctx, seed=7, max_cases=12, max_seconds=60.0, big=True)
    assert r["cases"] >= 3 and r["worst"] <= 5e-6, r


def test_fuzz_img_and_tiles(ctx):
    r = fuzzers.fuzz_img(ctx, seed=20261003, max_cases=240, max_seconds=90.0)
    assert r["cases"] >= 60 and r["lod_tiles"] >= 3, r


def test_fuzz_waveform_pyramid(ctx):
    r = fuzzers.fuzz_waveform(ctx, seed=20261003, max_cases=200, max_seconds=90.0)
    assert r["cases"] >= 50, r


def test_fuzz_track_manager_sessions(ctx):
    r = fuzzers.fuzz_track_manager(ctx, seed=20261003, max_cases=200, max_seconds=120.0)
    assert r["operations"] >= 40, r
