// tile_cache.cpp — see tile_cache.h.  Host code only (no device work): the C ABI below is usable without a GPU.
#include "tile_cache.h"

#include <algorithm>
#include <cstring>

#include "../../include/thesia_amd.h"
#include "common.h"

static uint64_t bump(uint64_t r) {  // wrapping_add(1).max(1), render_tiles.rs:88,93
    r += 1;
    return r < 1 ? 1 : r;
}

bool th_tile_cache::lookup(size_t id, uint32_t ch, uint32_t level, uint32_t tile_index, uint64_t *revision,
                           std::vector<uint8_t> *out) {
    std::lock_guard<std::mutex> lk(mu);
    if (revision) *revision = waveform_revision;
    auto it = entries.find(Key{id, ch, waveform_revision, level, tile_index});
    if (it == entries.end()) {
        misses++;
        return false;
    }
    lru.splice(lru.begin(), lru, it->second);  // last_used = next_tick(), :140
    if (out) *out = it->second->bytes;
    hits++;
    return true;
}

void th_tile_cache::store(size_t id, uint32_t ch, uint64_t revision, uint32_t level, uint32_t tile_index,
                          const uint8_t *src, size_t len) {
    std::lock_guard<std::mutex> lk(mu);
    if (revision != waveform_revision) return;  // :155-157
    const Key key{id, ch, revision, level, tile_index};
    auto it = entries.find(key);
    if (it != entries.end()) {  // replace without double counting, :192-200
        bytes -= std::min(bytes, it->second->bytes.size());
        lru.erase(it->second);
        entries.erase(it);
    }
    lru.push_front(Entry{key, std::vector<uint8_t>(src, src + len)});
    entries[key] = lru.begin();
    bytes += len;
    evict();
}

void th_tile_cache::evict() {
    while (bytes > budget_bytes && !lru.empty()) {
        const Entry &victim = lru.back();  // smallest last_used
        bytes -= victim.bytes.size();
        entries.erase(victim.key);
        lru.pop_back();
    }
}

void th_tile_cache::invalidate_waveform() {
    std::lock_guard<std::mutex> lk(mu);
    waveform_revision = bump(waveform_revision);
    lru.clear();  // clear_tiles, :220-224
    entries.clear();
    entries.rehash(0);
    bytes = 0;
}

void th_tile_cache::invalidate_spectrogram() {
    std::lock_guard<std::mutex> lk(mu);
    spectrogram_revision = bump(spectrogram_revision);
}

void th_tile_cache::set_budget(size_t budget) {
    std::lock_guard<std::mutex> lk(mu);
    budget_bytes = budget;
    evict();
}

// ------------------------------------------------------------------------------------------ C ABI
TH_API int th_tile_cache_create(size_t budget_bytes, th_tile_cache **out) {
    TH_TRY
    TH_REQUIRE(out, "out is NULL");
    *out = new th_tile_cache(budget_bytes ? budget_bytes : th_tile_cache::DEFAULT_BUDGET);
    return TH_OK;
    TH_CATCH
}

TH_API int th_tile_cache_destroy(th_tile_cache *c) {
    TH_TRY
    delete c;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tile_cache_lookup(th_tile_cache *c, size_t id, uint32_t ch, uint32_t level, uint32_t tile_index,
                                uint64_t *revision, uint8_t *out, size_t out_capacity, size_t *out_len, int *hit) {
    TH_TRY
    TH_REQUIRE(c && hit, "NULL argument");
    std::vector<uint8_t> v;
    const bool h = c->lookup(id, ch, level, tile_index, revision, &v);
    *hit = h ? 1 : 0;
    if (out_len) *out_len = h ? v.size() : 0;
    if (h) {
        if (!out || out_capacity < v.size()) return th::fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", v.size());
        if (!v.empty()) std::memcpy(out, v.data(), v.size());
    }
    return TH_OK;
    TH_CATCH
}

TH_API int th_tile_cache_store(th_tile_cache *c, size_t id, uint32_t ch, uint64_t revision, uint32_t level,
                               uint32_t tile_index, const uint8_t *bytes, size_t len) {
    TH_TRY
    TH_REQUIRE(c && (bytes || len == 0), "NULL argument");
    c->store(id, ch, revision, level, tile_index, bytes, len);
    return TH_OK;
    TH_CATCH
}

TH_API int th_tile_cache_invalidate(th_tile_cache *c, int waveform, int spectrogram) {
    TH_TRY
    TH_REQUIRE(c, "cache is NULL");
    if (waveform) c->invalidate_waveform();
    if (spectrogram) c->invalidate_spectrogram();
    return TH_OK;
    TH_CATCH
}

TH_API int th_tile_cache_set_budget(th_tile_cache *c, size_t budget_bytes) {
    TH_TRY
    TH_REQUIRE(c, "cache is NULL");
    c->set_budget(budget_bytes);
    return TH_OK;
    TH_CATCH
}

TH_API int th_tile_cache_stats(const th_tile_cache *c, size_t *entries, size_t *bytes, size_t *budget_bytes,
                               uint64_t *waveform_revision, uint64_t *spectrogram_revision, uint64_t *hits,
                               uint64_t *misses) {
    TH_TRY
    TH_REQUIRE(c, "cache is NULL");
    std::lock_guard<std::mutex> lk(c->mu);
    if (entries) *entries = c->entries.size();
    if (bytes) *bytes = c->bytes;
    if (budget_bytes) *budget_bytes = c->budget_bytes;
    if (waveform_revision) *waveform_revision = c->waveform_revision;
    if (spectrogram_revision) *spectrogram_revision = c->spectrogram_revision;
    if (hits) *hits = c->hits;
    if (misses) *misses = c->misses;
    return TH_OK;
    TH_CATCH
}
