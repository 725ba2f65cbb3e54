// tile_cache.h — host-side mirror of the reference's RenderTileCache (src-tauri/src/core/render_tiles.rs:51-230):
// a byte-budgeted least-recently-used cache of encoded waveform tiles keyed by
// (id, ch, waveform_revision, level, tile_index), plus the two revision counters the tile headers carry.
// The reference finds the eviction victim with a linear min_by_key scan over last_used ticks (:205-217); a
// recency list gives the same victim order in O(1).
#pragma once
#include <cstddef>
#include <cstdint>
#include <list>
#include <mutex>
#include <unordered_map>
#include <vector>

struct th_tile_cache {
    struct Key {
        size_t id;
        uint32_t ch;
        uint64_t revision;
        uint32_t level, tile_index;
        bool operator==(const Key &o) const {
            return id == o.id && ch == o.ch && revision == o.revision && level == o.level && tile_index == o.tile_index;
        }
    };
    struct KeyHash {
        size_t operator()(const Key &k) const {
            uint64_t h = 0x9E3779B97F4A7C15ull;
            for (uint64_t v : {(uint64_t)k.id, (uint64_t)k.ch, k.revision, (uint64_t)k.level, (uint64_t)k.tile_index}) {
                h ^= v + 0x9E3779B97F4A7C15ull + (h << 6) + (h >> 2);
            }
            return (size_t)h;
        }
    };
    struct Entry {
        Key key;
        std::vector<uint8_t> bytes;
    };
    static constexpr size_t DEFAULT_BUDGET = 32u * 1024u * 1024u;  // DEFAULT_WAVEFORM_CACHE_BUDGET_BYTES, :17

    explicit th_tile_cache(size_t budget) : budget_bytes(budget) {}

    // RenderTileCache::cached_waveform_tile, :124-144 — returns the current revision; on a hit copies the bytes
    // and makes the entry the most recently used one
    bool lookup(size_t id, uint32_t ch, uint32_t level, uint32_t tile_index, uint64_t *revision,
                std::vector<uint8_t> *out);
    // RenderTileCache::store_waveform_tile, :146-169 — dropped when `revision` is stale
    void store(size_t id, uint32_t ch, uint64_t revision, uint32_t level, uint32_t tile_index, const uint8_t *bytes,
               size_t len);
    void invalidate_waveform();     // :87-90: bump (never 0) and clear the tiles
    void invalidate_spectrogram();  // :92-94
    void invalidate_all() {         // :96-99
        invalidate_waveform();
        invalidate_spectrogram();
    }
    void set_budget(size_t budget);

    mutable std::mutex mu;  // the reference wraps the cache in an RwLock (lib.rs:38)
    std::list<Entry> lru;   // front = most recently used
    std::unordered_map<Key, std::list<Entry>::iterator, KeyHash> entries;
    size_t bytes = 0, budget_bytes;
    uint64_t waveform_revision = 1, spectrogram_revision = 1;
    uint64_t hits = 0, misses = 0;

  private:
    void evict();  // :205-218
};
