"""Randomised parity checks of the GPU path against the oracle (test infrastructure: the oracle is the checker).

Each fuzzer runs until `max_cases` cases or `max_seconds` have passed, from a fixed seed.  tests/test_gpu_fuzz.py runs
them inside `pytest -m gpu` (about 200 cases each); tests/fuzz_*.py and scripts/fuzz_stft.py are the long-running
command-line forms of the same loops."""
import os
import time

import numpy as np

import thesia_amd as ta
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cmap():
    return open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()


def fuzz_img(ctx, seed=1, max_cases=10 ** 9, max_seconds=30.0):
    """Random shapes / row ranges / dB ranges / colormap lengths through the quantiser, the level-0 tile encoder and the
    per-request LOD resize, bit for bit against the oracle (drawing.rs:4-33, render_tiles.rs:281-393)."""
    rng = np.random.default_rng(seed)
    cmap = _cmap()
    t_end = time.time() + max_seconds
    n_cases = n_lod = 0
    while time.time() < t_end and n_cases < max_cases:
        T, H = int(rng.integers(1, 1400)), int(rng.choice([1, 2, 40, 127, 128, 129, 130, 257, 513, 1025, int(rng.integers(1, 1200))]))
        spec = rng.uniform(-150, 20, (T, H)).astype(np.float32)
        k = max(1, spec.size // 500)
        spec.ravel()[rng.integers(0, spec.size, k)] = -np.inf
        spec.ravel()[rng.integers(0, spec.size, k)] = np.nan
        spec.ravel()[rng.integers(0, spec.size, k)] = np.inf
        lo = float(np.float32(rng.uniform(-140, -20)))
        hi = float(np.float32(lo + rng.choice([0.0, 1e-3, 1.0, 37.5, 60.0, 100.0, 159.99])))
        # quantisation boundaries: values whose scaled position is exactly k + 0.5
        cm = int(rng.choice([2, 3, 4, 16, 256, 258, 1024, 65536]))
        i0 = int(rng.integers(0, H))
        i1 = int(rng.integers(i0 + 1, H + int(rng.integers(1, 40))))  # rows >= H are zero-filled (mixed sample rates)
        got = ctx.spec_to_img(spec, (i0, i1), (lo, hi), cm)
        want = orc.convert_spectrogram_to_img(spec, (i0, i1), (lo, hi), cm)
        assert np.array_equal(got, want), (T, H, i0, i1, lo, hi, cm, int((got != want).sum()))
        if n_cases % 4 == 0 and got.shape[0] >= 1:
            img = got
            W_, H_ = img.shape[1], img.shape[0]
            tx, ty = int(rng.integers(0, -(-W_ // 512))), int(rng.integers(0, -(-H_ // 512)))
            a = ctx.encode_spectrogram_tile(img, cmap, 5, 0, 0, tx, ty)
            b = orc.encode_spectrogram_tile(img, cmap, 5, 0, 0, tx, ty)
            assert a == b, (W_, H_, tx, ty)
        if n_cases % 16 == 0 and got.shape[0] >= 2 and got.shape[1] >= 2:  # LOD > 0: the restated separable Lanczos3
            lx, ly = int(rng.integers(0, 4)), int(rng.integers(0, 4))
            if lx or ly:
                wl, hl = -(-got.shape[1] >> lx) if lx else got.shape[1], -(-got.shape[0] >> ly) if ly else got.shape[0]
                wl, hl = -(-got.shape[1] // (1 << lx)), -(-got.shape[0] // (1 << ly))
                tx, ty = int(rng.integers(0, -(-wl // 512))), int(rng.integers(0, -(-hl // 512)))
                a = ctx.encode_spectrogram_tile(got, cmap, 9, lx, ly, tx, ty)
                b = orc.encode_spectrogram_tile(got, cmap, 9, lx, ly, tx, ty)
                assert a == b, ("lod", got.shape, lx, ly, tx, ty)
                n_lod += 1
        n_cases += 1
    return {"cases": n_cases, "lod_tiles": n_lod}


def fuzz_waveform(ctx, seed=1, max_cases=10 ** 9, max_seconds=30.0):
    """Random channel lengths and level counts through th_waveform_pyramid_dev and the per-tile encoder against the oracle's
    encode_waveform_tile (render_tiles.rs:232-279): min / max bit-exact, mean bit-exact for bins <= 16 samples, 1e-6 of
    the peak above."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + max_seconds
    n_cases = 0
    while time.time() < t_end and n_cases < max_cases:
        n = int(rng.choice([int(rng.integers(1, 70)), int(rng.integers(1, 9000)), int(rng.integers(4000, 300000)),
                            4096 * int(rng.integers(1, 40)) + int(rng.integers(-2, 3))]))
        n = max(1, n)
        x = (rng.standard_normal(n) * rng.uniform(0.01, 0.5)).astype(np.float32)
        n_levels = int(rng.integers(1, 19))
        lv = ctx.waveform_pyramid(x, n_levels)
        peak = float(np.abs(x).max())
        for level in range(n_levels):
            bins = -(-n // (1 << level))
            assert lv[level].shape == (bins, 3), (n, level)
            n_tiles = -(-bins // 1024)
            for t in sorted({0, n_tiles - 1, int(rng.integers(0, n_tiles))}):
                want = np.frombuffer(orc.encode_waveform_tile(x, 1, level, t)[24:], np.float32).reshape(-1, 3)
                got = lv[level][1024 * t: 1024 * (t + 1)]
                assert got.shape == want.shape and np.array_equal(got[:, :2], want[:, :2]), (n, level, t)
                if level <= 4:
                    assert np.array_equal(got[:, 2], want[:, 2]), (n, level, t)
                else:
                    assert np.abs(got[:, 2] - want[:, 2]).max() <= 1e-6 * max(peak, 1e-30), (n, level, t)
                if n_cases % 8 == 0:  # the per-tile kernel as well (headers included)
                    assert ctx.encode_waveform_tile(x, 1, level, t)[:24] == orc.encode_waveform_tile(x, 1, level, t)[:24]
        n_cases += 1
    return {"cases": n_cases}


def fuzz_track_manager(ctx, seed=1, max_cases=10 ** 9, max_seconds=30.0):
    """Random sequences of TrackManager operations (add / remove / apply / set_setting / set_dB_range, mixed sample rates
    and channel counts) with the state checked against the oracle pipeline after every step (core/mod.rs:62-230).
    max_cases counts operations."""
    rng = np.random.default_rng(seed)
    cmap = _cmap()
    t_end = time.time() + max_seconds
    n_ops = n_runs = 0
    MAG_TOL = 1e-4
    while time.time() < t_end and n_ops < max_cases:
        tm = ta.TrackManager(ctx)
        tm.set_colormap(cmap)
        tracks, next_id = {}, 0
        setting, dB_range = (40.0, 4, 1, ta.MEL), 100.0
        for _ in range(int(rng.integers(3, 9))):
            op = rng.choice(["add", "add", "remove", "setting", "range"])
            if op == "add" or not tracks:
                new = []
                for _ in range(int(rng.integers(1, 3))):
                    sr = int(rng.choice([8000, 16000, 22050, 44100, 48000, 96000, 192000]))  # n_fft 512 .. 16384 with f_overlap 1 / 2
                    n = int(rng.integers(200, 30000 if sr <= 48000 else 90000))
                    wav = (rng.standard_normal((int(rng.integers(1, 3)), n)) * 0.1).astype(np.float32)
                    new.append((next_id, sr, wav))
                    tracks[next_id] = (sr, wav)
                    next_id += 1
                try:
                    tm.add_tracks(new)
                except ta.ThError as e:  # a sample rate whose plan the library refuses under the setting in force (mel filterbank above 1 GiB): transactional
                    assert e.code == -2, e
                    for tid_, _, _ in new:
                        del tracks[tid_]
                tm.apply_track_list_changes()
            elif op == "remove":
                tid = int(rng.choice(list(tracks)))
                tm.remove_track(tid)
                del tracks[tid]
                tm.apply_track_list_changes()
            elif op == "setting":
                prev_setting = setting
                setting = (float(rng.choice([20.0, 40.0, 2048 / 48])), int(rng.choice([2, 4, 8, 16, 32])), int(rng.choice([1, 2])),
                           int(rng.choice([ta.MEL, ta.LINEAR])))
                if rng.random() < 0.25:  # the ends of what the UI accepts (winMillisec has a lower bound only): n_fft 4 .. 65536
                    setting = (float(rng.choice([1.0, 2.5, 170.0, 400.0])), int(rng.choice([1, 2, 4])), int(rng.choice([1, 2])), ta.LINEAR)
                elif rng.random() < 0.2:  # Mel under long windows (round 6: the moment-form epilogues of n_fft 4096 / 8192 / 16384) and an f_overlap that is no power of two
                    setting = (float(rng.choice([85.0, 170.0, 340.0])), int(rng.choice([2, 4])), int(rng.choice([1, 1, 3])), ta.MEL)
                try:
                    tm.set_setting(*setting)
                except ta.ThError as e:  # a setting the library refuses (mel filterbank above 1 GiB under a long window at a high rate): nothing may have changed
                    assert e.code == -2, e
                    setting = prev_setting
            else:
                dB_range = float(rng.choice([40.0, 60.0, 100.0, 120.0]))
                tm.set_dB_range(dB_range)
            n_ops += 1
            if not tracks:
                continue
            # ---- the oracle's view of the state
            max_sr = max(sr for sr, _ in tracks.values())
            mins, maxs, specs = [], [], {}
            for tid, (sr, wav) in tracks.items():
                hop, win, n_fft = orc.calc_framing_params(setting[0], setting[1], setting[2], sr)
                fb = orc.calc_mel_fb_default(sr, n_fft) if setting[3] == ta.MEL else None
                for ch in range(wav.shape[0]):
                    want = orc.calc_spec(wav[ch], win, hop, n_fft, mel_fb=fb)
                    got = tm.spec(tid, ch)
                    assert got.shape == want.shape, (setting, sr, got.shape, want.shape)
                    a, b = np.power(10.0, got.astype(np.float64) / 20), np.power(10.0, want.astype(np.float64) / 20)
                    scale = np.maximum(b.max(axis=1, keepdims=True), 1e-30)
                    assert (np.abs(a - b) / scale).max() <= MAG_TOL, (setting, sr)
                    specs[(tid, ch)] = (got, sr)
                    mins.append(got.min())
                    maxs.append(got.max())
            lo, hi = orc.global_db_range(mins, maxs, dB_range)
            glo, ghi, gsr = tm.db_state()
            assert (glo, ghi, gsr) == (lo, hi, max_sr), ((glo, ghi, gsr), (lo, hi, max_sr))
            for (tid, ch), (got, sr) in specs.items():
                r = orc.hz_range_to_idx(orc.MEL if setting[3] == ta.MEL else orc.LINEAR, (0.0, max_sr / 2), sr, got.shape[1])
                img = tm.img(tid, ch)
                assert np.array_equal(img, orc.convert_spectrogram_to_img(got, r, (lo, hi), 258)), (setting, sr, r)
                _, srev = tm.revisions()
                assert tm.get_spectrogram_tile(tid, ch, 0, 0, 0, 0) == orc.encode_spectrogram_tile(img, cmap, srev, 0, 0, 0, 0)
            # waveform tiles of random levels (0 and 1 come from the samples, 2 .. from the resident pyramid, levels above the
            # last one are a single bin) against the oracle's encode_waveform_tile
            wrev, _ = tm.revisions()
            for _ in range(3):
                tid = int(rng.choice(list(tracks)))
                sr_, wav_ = tracks[tid]
                ch_ = int(rng.integers(0, wav_.shape[0]))
                level = int(rng.integers(0, 17))
                n_t = -(-(-(-wav_.shape[1] // (1 << level))) // 1024)
                t_ = int(rng.integers(0, n_t + 1))
                got_t, want_t = tm.get_waveform_tile(tid, ch_, level, t_), orc.encode_waveform_tile(wav_[ch_], wrev, level, t_)
                assert got_t[:24] == want_t[:24], (tid, ch_, level, t_)
                gb, wb = np.frombuffer(got_t[24:], np.float32).reshape(-1, 3), np.frombuffer(want_t[24:], np.float32).reshape(-1, 3)
                assert gb.shape == wb.shape and np.array_equal(gb[:, :2], wb[:, :2]), (tid, ch_, level, t_)
                if gb.size:
                    tol = 0.0 if level <= 4 else 1e-6 * float(np.abs(wav_[ch_]).max())
                    assert np.abs(gb[:, 2] - wb[:, 2]).max() <= tol, (tid, ch_, level, t_)
            # the batched fetch answers a random mix of requests (level 0, LOD, past-the-end tiles) like the single-tile call
            keys = list(specs)
            reqs = [(*keys[int(rng.integers(0, len(keys)))], int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 3)), 0)
                    for _ in range(5)]
            assert tm.get_spectrogram_tiles(reqs, pinned=bool(rng.integers(0, 2))) == [tm.get_spectrogram_tile(*r) for r in reqs]
        tm.close()
        n_runs += 1
    return {"sessions": n_runs, "operations": n_ops}


def fuzz_stft(ctx, seed=1, max_cases=10 ** 9, max_seconds=60.0, big=False, cap_bytes=None):
    """Random framings / lengths / batch shapes through the auto-selected kernel (the wave kernels with all their modes —
    register reuse, phased, dynamic, boundary frames, fused mel, several frames per wave — the block kernel of n_fft 8192 /
    16384, the matrix-core mel path) against the generic kernel, which shares none of that code."""
    rng = np.random.default_rng(seed)
    t_end = time.time() + max_seconds
    n_cases = 0
    worst = 0.0
    while time.time() < t_end and n_cases < max_cases:
        n_fft = int(rng.choice([512, 1024, 2048, 2048, 4096, 8192, 16384, 32768]))  # multi-frame, one-frame, block and subwave kernels
        if big and rng.random() < 0.04:
            n_fft = 65536  # (round 5: stft_subwave2_kernel; rare — the generic kernel it is checked against takes its time there)
        win = int(rng.integers(n_fft // 2 + 1, n_fft + 1))
        if rng.random() < 0.5:
            win = n_fft if rng.random() < 0.5 else win // 2 * 2
        hop = int(rng.choice([win // 4, win // 2, win // 8, win // 16, win // 32, int(rng.integers(1, win + 1)), int(rng.integers(1, 300)),
                              480, 441, 160, 221, 320, 240, 120, 80, 60]))
        hop = max(1, min(hop, win))
        mel = rng.random() < 0.35
        n_mel = int(rng.choice([0, 40, 128, 200])) if mel else 0
        sr = int(rng.choice([16000, 22050, 44100, 48000])) if n_fft <= 4096 else int(rng.choice([48000, 96000, 192000]))
        if n_fft == 512 and mel:  # the banded-sum mel kernel: default mel counts of 8-12 kHz audio, and 512 mels
            sr, n_mel = int(rng.choice([8000, 11025, 12000, sr])), int(rng.choice([0, 0, 512, n_mel]))
        if n_fft == 4096 and rng.random() < 0.5:  # the 88.2 / 96 kHz framings (grid-aligned reuse: 8 waves and one window table for even hops, 7 and two for 3528 / 441)
            sr = int(rng.choice([88200, 96000]))
            win = sr // 25
            hop = win // int(rng.choice([2, 4, 4, 8, 16, 32]))
        try:
            plan = ta.Plan(ctx, sr, win, hop, n_fft, ta.MEL if mel else ta.LINEAR, n_mel)
            ref = ta.Plan(ctx, sr, win, hop, n_fft, ta.MEL if mel else ta.LINEAR, n_mel)
        except ta.ThError:
            continue
        ref.set_kernel(1)
        lens = [int(rng.integers(1, 6 * n_fft)) for _ in range(int(rng.integers(1, 5)))] + [int(rng.integers(n_fft, min(40 * n_fft, 200000)))]
        if big:
            lens += [int(rng.integers(min(200 * n_fft, 1000000), min(1500 * n_fft, 4000000))) for _ in range(int(rng.integers(1, 4)))]
            hop = max(hop, 64)
        if hop < 8:
            lens = [min(v, 3 * n_fft) for v in lens]
        elif hop < 64:
            lens = [min(v, 20 * n_fft) for v in lens]
        if cap_bytes is not None:
            # soak runs (scripts/fuzz_soak.py, open-ended seeds): bound one case's spectrograms — a long channel at a small hop and
            # n_fft 32768 is 3-4 GB of f32 per plan, and the checker below holds two plans' worth plus float64 temporaries on the host
            h = plan.height
            while sum(v // hop + 1 for v in lens) * h * 4 > cap_bytes and max(lens) > 2 * n_fft:
                lens[int(np.argmax(lens))] //= 2
        wavs = [(rng.standard_normal(v) * 0.1 + 0.3 * np.sin(np.arange(v) * rng.uniform(0.001, 1.0))).astype(np.float32) for v in lens]
        a, mma = plan.calc_spec_batch(wavs)
        b, mmb = ref.calc_spec_batch(wavs)
        for i, (x, y) in enumerate(zip(a, b)):
            assert x.shape == y.shape, (win, hop, n_fft, lens[i])
            ax, ay = np.power(10.0, x.astype(np.float64) / 20), np.power(10.0, y.astype(np.float64) / 20)
            scale = np.maximum(ay.max(axis=1, keepdims=True), 1e-30)
            err = float((np.abs(ax - ay) / scale).max()) if x.size else 0.0
            worst = max(worst, err)
            assert err <= 5e-6, (plan.kernel_name, win, hop, n_fft, n_mel, lens[i], err)
            assert mma[i, 0] == x.min() and mma[i, 1] == x.max(), (plan.kernel_name, win, hop, n_fft, lens[i])
        plan.close()
        ref.close()
        n_cases += 1
    return {"cases": n_cases, "worst": worst}
