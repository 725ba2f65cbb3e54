"""Waveform-tile LRU cache (th_tile_cache, host only) against the reference's own cache tests
(src-tauri/src/core/render_tiles.rs:473-537) and against the literal Python restatement in oracle/ on random
operation sequences."""
import numpy as np
import pytest

import thesia_amd as ta
from oracle import oracle as orc

TILE = 24 + 1024 * 12  # header + WAVEFORM_TILE_BINS x 3 f32 (render_tiles.rs:14,232-259)


def _tile(seed: int, n: int = TILE) -> bytes:
    return np.random.default_rng(seed).integers(0, 256, n, dtype=np.uint8).tobytes()


def _wf(n_tiles: int, revision: int, tile_index: int) -> bytes:
    """encode_waveform_tile(&vec![0.0; WAVEFORM_TILE_BINS * n_tiles], revision, 0, tile_index) as in the reference's
    cache tests"""
    return orc.encode_waveform_tile(np.zeros(1024 * n_tiles, np.float32), revision, 0, tile_index)


def test_cache_evicts_and_invalidates():
    # render_tiles.rs:473-487
    with ta.TileCache(TILE) as c:
        rev = c.stats()["waveform_revision"]
        c.store(1, 0, rev, 0, 0, _wf(2, rev, 0))
        c.store(1, 0, rev, 0, 1, _wf(2, rev, 1))
        st = c.stats()
        assert st["entries"] == 1 and st["bytes"] <= st["budget_bytes"]
        c.invalidate_waveform()
        st = c.stats()
        assert st["entries"] == 0 and st["bytes"] == 0 and st["waveform_revision"] > rev


def test_cache_replaces_duplicate_without_double_counting():
    # render_tiles.rs:489-501
    with ta.TileCache(2**40) as c:
        rev = c.stats()["waveform_revision"]
        b = _wf(1, rev, 0)
        assert len(b) == TILE
        c.store(1, 0, rev, 0, 0, b)
        c.store(1, 0, rev, 0, 0, b)
        st = c.stats()
        assert st["entries"] == 1 and st["bytes"] == len(b)


def test_cache_hit_updates_lru_order():
    # render_tiles.rs:503-523
    with ta.TileCache(TILE * 2) as c:
        rev = c.stats()["waveform_revision"]
        for i in range(2):
            c.store(1, 0, rev, 0, i, _wf(3, rev, i))
        assert c.lookup(1, 0, 0, 0)[1] is not None  # touch tile 0: tile 1 is now the oldest
        c.store(1, 0, rev, 0, 2, _wf(3, rev, 2))
        assert c.lookup(1, 0, 0, 0)[1] == _wf(3, rev, 0)
        assert c.lookup(1, 0, 0, 1)[1] is None
        assert c.lookup(1, 0, 0, 2)[1] == _wf(3, rev, 2)


def test_cache_drops_tile_from_stale_revision():
    # render_tiles.rs:525-537
    with ta.TileCache() as c:
        st = c.stats()
        assert st["budget_bytes"] == 32 * 1024 * 1024  # DEFAULT_WAVEFORM_CACHE_BUDGET_BYTES, :17
        rev = st["waveform_revision"]
        c.invalidate_waveform()
        c.store(1, 0, rev, 0, 0, _wf(1, rev, 0))
        assert c.stats()["entries"] == 0


def test_revisions_never_zero_and_spectrogram_invalidation_keeps_tiles():
    with ta.TileCache() as c:
        rev = c.stats()["waveform_revision"]
        c.store(7, 1, rev, 3, 5, _tile(4, 100))
        c.invalidate_spectrogram()  # render_tiles.rs:92-94: bumps only the spectrogram revision
        st = c.stats()
        assert st["spectrogram_revision"] == 2 and st["waveform_revision"] == rev and st["entries"] == 1
        c.invalidate_all()
        st = c.stats()
        assert st["spectrogram_revision"] == 3 and st["waveform_revision"] == rev + 1 and st["entries"] == 0


def test_oversized_tile_is_not_kept_and_empty_tile_is():
    with ta.TileCache(100) as c:
        rev = c.stats()["waveform_revision"]
        c.store(1, 0, rev, 0, 0, _tile(5, 101))  # inserted, then evicted by the budget loop (:205-218)
        assert c.stats()["entries"] == 0 and c.stats()["bytes"] == 0
        c.store(1, 0, rev, 0, 1, b"")
        assert c.lookup(1, 0, 0, 1)[1] == b""


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_sequences_match_the_restatement(seed):
    rng = np.random.default_rng(seed)
    budget = int(rng.integers(2_000, 40_000))
    ref = orc.RenderTileCache(budget)
    with ta.TileCache(budget) as c:
        stale = [1]
        for step in range(3000):
            op = rng.random()
            key = (int(rng.integers(0, 3)), int(rng.integers(0, 2)), int(rng.integers(0, 4)), int(rng.integers(0, 6)))
            if op < 0.45:
                data = _tile(step, int(rng.integers(0, 3000)))
                rev = ref.waveform_revision if rng.random() < 0.9 else int(rng.choice(stale))
                ref.store_waveform_tile(key[0], key[1], rev, key[2], key[3], data)
                c.store(key[0], key[1], rev, key[2], key[3], data)
            elif op < 0.9:
                r0, d0 = ref.cached_waveform_tile(*key)
                r1, d1 = c.lookup(*key)
                assert r0 == r1 and d0 == d1, (step, key)
            elif op < 0.95:
                stale.append(ref.waveform_revision)
                ref.invalidate_waveform()
                c.invalidate_waveform()
            elif op < 0.98:
                ref.invalidate_spectrogram()
                c.invalidate_spectrogram()
            else:
                budget = int(rng.integers(1_000, 40_000))
                ref.budget_bytes = budget
                ref._evict()
                c.set_budget(budget)
            st = c.stats()
            assert (st["entries"], st["bytes"], st["waveform_revision"], st["spectrogram_revision"]) == (
                len(ref.entries), ref.bytes, ref.waveform_revision, ref.spectrogram_revision), step
        # every surviving entry is byte-identical
        for (i, ch, rev, lv, ti), (data, _) in list(ref.entries.items()):
            assert c.lookup(i, ch, lv, ti)[1] == data
