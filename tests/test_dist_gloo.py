"""N > 1 path on CPU: world_size-2 gloo processes shard the tracks (th_shard_assign), compute
their shard (here with the CPU oracle — the GPU kernels are covered by the -m gpu tests),
exchange the global dB range with the path's single 2-float all-reduce, quantise, encode tiles
and gather them at the root.  The result must be identical to the single-process pipeline."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _pipeline_inputs():
    from tests.synth import synth_track
    sr, win, hop, n_fft = 48000, 2048, 512, 2048
    lens = [9000, 30000, 12000, 5000, 30000, 7000, 16000]
    return sr, win, hop, n_fft, [synth_track(i, sr, n) for i, n in enumerate(lens)]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    import thesia_amd as ta
    from thesia_amd import dist as tdist
    from oracle import oracle as orc
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sr, win, hop, n_fft, wavs = _pipeline_inputs()
        weights = [ta.stft_n_frames(len(w), win, hop) for w in wavs]
        owner = ta.shard_assign(weights, world)
        mine = [i for i in range(len(wavs)) if owner[i] == rank]
        specs = {i: orc.calc_spec(wavs[i], win, hop, n_fft) for i in mine}
        lmin = min((float(s.min()) for s in specs.values()), default=np.inf)
        lmax = max((float(s.max()) for s in specs.values()), default=-np.inf)
        gmin, gmax = tdist.allreduce_min_max(lmin, lmax, dist)
        lo, hi = ta.global_db_range([gmin], [gmax], 100.0)
        cmap = open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
        tiles = []
        for i in mine:
            img = orc.convert_spectrogram_to_img(specs[i], (0, specs[i].shape[1]), (lo, hi), 258)
            tiles.append(i.to_bytes(4, "little") + orc.encode_spectrogram_tile(img, cmap, 1, 0, 0, 0, 1))
        got = tdist.gather_bytes_to_root(tiles, dist, root=0)
        # the device-resident form (on GPUs: the RGBA buffers themselves; here CPU tensors over gloo): one tensor per rank
        import torch
        mine_t = torch.frombuffer(bytearray(b"".join(tiles)) or bytearray(1), dtype=torch.uint8)[: sum(map(len, tiles))]
        got_t = tdist.gather_tensor_to_root(mine_t, dist, root=0)
        if got_t is not None:
            got_t = [t.numpy().tobytes() for t in got_t]
        q.put((rank, owner.tolist(), (lo, hi), got, got_t))
    finally:
        dist.destroy_process_group()


def test_two_rank_pipeline_matches_single_process():
    import torch.multiprocessing as mp
    import thesia_amd as ta
    from oracle import oracle as orc
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    res.sort(key=lambda r: r[0])
    # single-process reference
    sr, win, hop, n_fft, wavs = _pipeline_inputs()
    specs = [orc.calc_spec(w, win, hop, n_fft) for w in wavs]
    lo, hi = orc.global_db_range([s.min() for s in specs], [s.max() for s in specs], 100.0)
    cmap = open(os.path.join(ROOT, "tests", "golden", "colormap_inferno_rgba258.bin"), "rb").read()
    want = {i: orc.encode_spectrogram_tile(orc.convert_spectrogram_to_img(s, (0, s.shape[1]), (lo, hi), 258), cmap, 1,
                                           0, 0, 0, 1) for i, s in enumerate(specs)}
    owner = res[0][1]
    assert owner == res[1][1] and set(owner) == {0, 1}
    assert res[0][2] == res[1][2] == (lo, hi)          # both ranks agree on the global dB range
    assert res[1][3] is None                           # only the root receives the gather
    # the tensor gather delivers the same bytes, rank by rank
    assert res[1][4] is None and res[0][4] == [b"".join(parts) for parts in res[0][3]]
    gathered = {}
    for r, parts in enumerate(res[0][3]):
        for b in parts:
            i = int.from_bytes(b[:4], "little")
            assert owner[i] == r
            gathered[i] = b[4:]
    assert gathered == want


def test_shard_assign_is_balanced_and_deterministic():
    import thesia_amd as ta
    eq = ta.shard_assign([2813] * 1024, 8)
    assert np.bincount(eq, minlength=8).tolist() == [128] * 8           # BASELINE config 5: 128 tracks per GPU
    assert eq[:16].tolist() == [0, 1, 2, 3, 4, 5, 6, 7] * 2             # equal weights: round-robin
    rng = np.random.default_rng(0)
    w = rng.integers(100, 6000, 200)
    own = ta.shard_assign(w, 8)
    loads = np.bincount(own, weights=w, minlength=8)
    assert loads.max() - loads.min() <= w.max()
    assert np.array_equal(own, ta.shard_assign(w, 8))
    assert ta.shard_assign([], 4).size == 0 and ta.shard_assign([5, 1], 1).tolist() == [0, 0]
