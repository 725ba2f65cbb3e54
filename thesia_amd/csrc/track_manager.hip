// track_manager.hip — C++ mirror of the reference's TrackManager orchestration
// (src-tauri/src/core/mod.rs:33-230) and of the tile commands that read it
// (src-tauri/src/lib.rs:342-389), with audio / f32 dB specs / u16 images resident in HBM.
//
// What is mirrored:  add_tracks → update_specs (mod.rs:62-71,137-164), remove_tracks (:84-100),
// apply_track_list_changes → update_spec_imgs (:102-105,168-230) incl. no_spec_img_ids /
// need_update_all, set_setting (:107-115), set_dB_range (:123-126), set_colormap_length (:128-131),
// the revision bumps the command layer applies around them (lib.rs:192,221,244,265,284), and
// get_waveform_tile / get_spectrogram_tile.
// What is not: file decoding, normalisation / clip guarding (TrackList, out of scope — the path
// starts at an in-memory planar f32 channel, audio.rs:65-78).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <map>
#include <set>
#include <tuple>
#include <vector>

#include "common.h"
#include "context.h"
#include "host_math.h"
#include "kernels.h"
#include "tile_cache.h"

using namespace th;

namespace {

struct Channel {
    float *d_wav = nullptr;
    size_t n = 0;
    float *d_spec = nullptr;
    size_t T = 0, H = 0, spec_pitch = 0;   // rows padded to 128 B (th_pitch_f32)
    uint16_t *d_img = nullptr;
    size_t img_h = 0, img_w = 0, img_pitch = 0;  // rows padded to 128 B (th_pitch_u16)
    float mn = INFINITY, mx = -INFINITY;  // find_min_max of this spec (simd.rs:14-36)
    bool has_spec = false;
};

struct Track {
    uint32_t sr = 0;
    std::vector<Channel> ch;
};

using PlanKey = std::tuple<uint32_t, size_t, size_t, size_t, int>;  // sr, win, hop, n_fft, scale

}  // namespace

struct th_tm {
    th_ctx *ctx = nullptr;
    // SpecSetting::new — spectrogram.rs:47-54
    double win_ms = 40.;
    uint32_t t_overlap = 4, f_overlap = 1;
    int freq_scale = TH_FREQ_MEL;
    // TrackManager::new — core/mod.rs:46-60
    float max_dB = -INFINITY, min_dB = INFINITY;
    uint32_t max_sr = 0;
    float dB_range = 100.f;
    uint32_t colormap_length = 258;
    std::vector<size_t> no_spec_img_ids;
    std::map<size_t, Track> tracks;
    std::map<PlanKey, th_plan *> plans;  // SpectrogramAnalyzer caches, spectrogram.rs:101-105
    // RenderTileCache state that the tile encoders need — render_tiles.rs:68-96
    std::vector<uint8_t> colormap_rgba{0, 0, 0, 255, 255, 255, 255, 255};
    th_tile_cache cache{th_tile_cache::DEFAULT_BUDGET};  // revisions + waveform-tile LRU, render_tiles.rs:51-230

    void invalidate_waveform() { cache.invalidate_waveform(); }
    void invalidate_spectrogram() { cache.invalidate_spectrogram(); }
    void invalidate_all() { cache.invalidate_all(); }
    uint64_t waveform_revision() const {
        std::lock_guard<std::mutex> lk(cache.mu);
        return cache.waveform_revision;
    }
    uint64_t spectrogram_revision() const {
        std::lock_guard<std::mutex> lk(cache.mu);
        return cache.spectrogram_revision;
    }
};

namespace {

void free_channel(Channel &c) {
    if (c.d_wav) (void)hipFree(c.d_wav);
    if (c.d_spec) (void)hipFree(c.d_spec);
    if (c.d_img) (void)hipFree(c.d_img);
    c = Channel();
}

int get_plan(th_tm *tm, uint32_t sr, th_plan **out) {
    size_t hop, win, n_fft;
    calc_framing_params(tm->win_ms, tm->t_overlap, tm->f_overlap, sr, &hop, &win, &n_fft);
    const PlanKey key{sr, win, hop, n_fft, tm->freq_scale};
    auto it = tm->plans.find(key);
    if (it != tm->plans.end()) {
        *out = it->second;
        return TH_OK;
    }
    th_plan *p = nullptr;
    int rc = th_plan_create(tm->ctx, sr, win, hop, n_fft, tm->freq_scale, 0, &p);
    if (rc != TH_OK) return rc;
    tm->plans[key] = p;
    *out = p;
    return TH_OK;
}

// SpectrogramAnalyzer::retain — spectrogram.rs:156-185: drop plans no resident track needs
void retain_plans(th_tm *tm) {
    std::set<PlanKey> need;
    for (auto &kv : tm->tracks) {
        size_t hop, win, n_fft;
        calc_framing_params(tm->win_ms, tm->t_overlap, tm->f_overlap, kv.second.sr, &hop, &win, &n_fft);
        need.insert(PlanKey{kv.second.sr, win, hop, n_fft, tm->freq_scale});
    }
    for (auto it = tm->plans.begin(); it != tm->plans.end();) {
        if (!need.count(it->first)) {
            th_plan_destroy(it->second);
            it = tm->plans.erase(it);
        } else {
            ++it;
        }
    }
}

// TrackManager::update_specs — core/mod.rs:137-164: one batched launch per plan
int update_specs(th_tm *tm, const std::vector<size_t> &ids) {
    th_ctx *c = tm->ctx;
    std::map<th_plan *, std::vector<Channel *>> groups;
    for (size_t id : ids) {
        auto it = tm->tracks.find(id);
        if (it == tm->tracks.end()) continue;
        th_plan *p = nullptr;
        int rc = get_plan(tm, it->second.sr, &p);
        if (rc != TH_OK) return rc;
        for (Channel &ch : it->second.ch) groups[p].push_back(&ch);
    }
    for (auto &kv : groups) {
        th_plan *p = kv.first;
        std::vector<Channel *> &chs = kv.second;
        std::vector<th_chan_desc> descs(chs.size());
        for (size_t i = 0; i < chs.size(); i++) {
            Channel &ch = *chs[i];
            const size_t T = stft_n_frames(ch.n, p->g.win, p->g.hop), H = p->g.height;
            if (ch.d_spec && (ch.T != T || ch.H != H)) {
                TH_HIP(hipFree(ch.d_spec));
                ch.d_spec = nullptr;
            }
            const size_t pitch = th_pitch_f32(H);
            if (!ch.d_spec) TH_HIP(hipMalloc((void **)&ch.d_spec, std::max<size_t>(1, T * pitch) * sizeof(float)));
            ch.T = T;
            ch.H = H;
            ch.spec_pitch = pitch;
            descs[i] = th_chan_desc{ch.d_wav, ch.d_spec, ch.n, T, pitch};
        }
        float *d_mm = nullptr;
        TH_HIP(hipMalloc((void **)&d_mm, 2 * chs.size() * sizeof(float)));
        int rc = th_calc_spec_batch_dev(p, descs.data(), descs.size(), d_mm);
        std::vector<float> mm(2 * chs.size());
        hipError_t e = hipSuccess;
        if (rc == TH_OK) {
            e = hipMemcpyAsync(mm.data(), d_mm, mm.size() * sizeof(float), hipMemcpyDeviceToHost, c->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        }
        (void)hipFree(d_mm);
        if (rc != TH_OK) return rc;
        TH_HIP(e);
        for (size_t i = 0; i < chs.size(); i++) {
            chs[i]->mn = mm[2 * i];
            chs[i]->mx = mm[2 * i + 1];
            chs[i]->has_spec = true;
        }
    }
    return TH_OK;
}

// TrackManager::update_spec_imgs — core/mod.rs:168-230
int update_spec_imgs(th_tm *tm, bool force_update_all, std::vector<size_t> *updated) {
    float mn = INFINITY, mx = -INFINITY;
    {
        std::vector<float> mins, maxs;
        for (auto &kv : tm->tracks)
            for (Channel &ch : kv.second.ch)
                if (ch.has_spec) {
                    mins.push_back(ch.mn);
                    maxs.push_back(ch.mx);
                }
        global_db_range(mins.data(), maxs.data(), mins.size(), tm->dB_range, &mn, &mx);  // :169-180
    }
    bool need_update_all = force_update_all;
    if (!(tm->max_dB == mx)) {  // `!=` on f32 (NaN never equal)  :182-185
        tm->max_dB = mx;
        need_update_all = true;
    }
    if (!(tm->min_dB == mn)) {
        tm->min_dB = mn;
        need_update_all = true;
    }
    uint32_t max_sr = 0;  // TrackList::max_sr, track.rs:371-376
    for (auto &kv : tm->tracks) max_sr = std::max(max_sr, kv.second.sr);
    if (tm->max_sr != max_sr) {
        tm->max_sr = max_sr;
        need_update_all = true;
    }
    std::set<size_t> ids;
    if (need_update_all) {
        tm->no_spec_img_ids.clear();
        for (auto &kv : tm->tracks) ids.insert(kv.first);
    } else {
        ids.insert(tm->no_spec_img_ids.begin(), tm->no_spec_img_ids.end());
        tm->no_spec_img_ids.clear();
    }
    updated->assign(ids.begin(), ids.end());
    if (ids.empty()) return TH_OK;

    if (need_update_all)  // spec_imgs.clear()  :224-226
        for (auto &kv : tm->tracks)
            for (Channel &ch : kv.second.ch)
                if (ch.d_img && !ids.count(kv.first)) {
                    TH_HIP(hipFree(ch.d_img));
                    ch.d_img = nullptr;
                    ch.img_h = ch.img_w = 0;
                }
    std::vector<th_img_desc> descs;
    for (size_t id : ids) {
        auto it = tm->tracks.find(id);
        if (it == tm->tracks.end()) continue;  // filter over specs: ids without a spec are skipped
        for (Channel &ch : it->second.ch) {
            if (!ch.has_spec) continue;
            size_t i0, i1;
            hz_range_to_idx(tm->freq_scale, 0.f, float(tm->max_sr) / 2.f, it->second.sr, ch.H, &i0, &i1);  // :210-214
            const size_t h = i1 - i0, w = ch.T;
            if (ch.d_img && (ch.img_h != h || ch.img_w != w)) {
                TH_HIP(hipFree(ch.d_img));
                ch.d_img = nullptr;
            }
            const size_t ipitch = th_pitch_u16(w);
            if (!ch.d_img) TH_HIP(hipMalloc((void **)&ch.d_img, std::max<size_t>(1, h * ipitch) * sizeof(uint16_t)));
            ch.img_h = h;
            ch.img_w = w;
            ch.img_pitch = ipitch;
            descs.push_back(th_img_desc{ch.d_spec, ch.d_img, ch.T, ch.H, i0, i1, ch.spec_pitch, ipitch});
        }
    }
    return th_spec_to_img_batch_dev(tm->ctx, descs.data(), descs.size(), tm->min_dB, tm->max_dB, tm->colormap_length);
}

Channel *find_channel(th_tm *tm, size_t id, uint32_t ch) {
    auto it = tm->tracks.find(id);
    if (it == tm->tracks.end() || ch >= it->second.ch.size()) return nullptr;
    return &it->second.ch[ch];
}

}  // namespace

TH_API int th_tm_create(th_ctx *c, th_tm **out) {
    TH_TRY
    TH_REQUIRE(c && out, "NULL argument");
    th_tm *tm = new th_tm();
    tm->ctx = c;
    *out = tm;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_destroy(th_tm *tm) {
    TH_TRY
    if (!tm) return TH_OK;
    (void)hipSetDevice(tm->ctx->device);
    (void)hipStreamSynchronize(tm->ctx->stream);
    for (auto &kv : tm->tracks)
        for (Channel &ch : kv.second.ch) free_channel(ch);
    for (auto &kv : tm->plans) th_plan_destroy(kv.second);
    delete tm;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_set_colormap(th_tm *tm, const uint8_t *rgba, size_t bytes) {
    TH_TRY
    TH_REQUIRE(tm && rgba, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    // RenderTileCache::set_colormap keeps the old map for malformed input — render_tiles.rs:80-85
    if (bytes >= 4 && bytes % 4 == 0) tm->colormap_rgba.assign(rgba, rgba + bytes);
    tm->invalidate_spectrogram();
    // init: TM.set_colormap_length(len / 4) → update_spec_imgs(force) — lib.rs:61, core/mod.rs:128-131
    tm->colormap_length = (uint32_t)(bytes / 4);
    std::vector<size_t> upd;
    return update_spec_imgs(tm, true, &upd);
    TH_CATCH
}

TH_API int th_tm_set_setting(th_tm *tm, double win_ms, uint32_t t_overlap, uint32_t f_overlap, int freq_scale) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    TH_REQUIRE(win_ms > 0. && t_overlap >= 1 && f_overlap >= 1, "invalid SpecSetting (lib.rs:275-277)");
    TH_REQUIRE(freq_scale == TH_FREQ_LINEAR || freq_scale == TH_FREQ_MEL, "bad freq_scale");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    tm->win_ms = win_ms;
    tm->t_overlap = t_overlap;
    tm->f_overlap = f_overlap;
    tm->freq_scale = freq_scale;
    retain_plans(tm);  // spec_analyzer.retain(...)  core/mod.rs:111-112
    std::vector<size_t> ids;
    for (auto &kv : tm->tracks) ids.push_back(kv.first);
    int rc = update_specs(tm, ids);
    if (rc != TH_OK) return rc;
    std::vector<size_t> upd;
    rc = update_spec_imgs(tm, true, &upd);
    tm->invalidate_spectrogram();  // lib.rs:284
    return rc;
    TH_CATCH
}

TH_API int th_tm_set_dB_range(th_tm *tm, float dB_range) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    TH_REQUIRE(dB_range > 0.f, "dB_range must be > 0 (lib.rs:259)");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    tm->dB_range = dB_range;
    std::vector<size_t> upd;
    int rc = update_spec_imgs(tm, true, &upd);
    tm->invalidate_spectrogram();  // lib.rs:265
    return rc;
    TH_CATCH
}

TH_API int th_tm_add_tracks(th_tm *tm, size_t n_tracks, const size_t *ids, const uint32_t *srs,
                            const uint32_t *n_channels, const float *const *channels_flat, const size_t *n_samples) {
    TH_TRY
    TH_REQUIRE(tm && ids && srs && n_channels && channels_flat && n_samples, "NULL argument");
    TH_REQUIRE(n_tracks >= 1, "no tracks (lib.rs:182)");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    th_ctx *c = tm->ctx;
    TH_HIP(hipSetDevice(c->device));
    size_t flat = 0;
    std::vector<size_t> added;
    for (size_t t = 0; t < n_tracks; t++) {
        TH_REQUIRE(srs[t] > 0 && n_channels[t] >= 1 && n_samples[t] >= 1, "track %zu: empty or invalid", ids[t]);
        Track &tr = tm->tracks[ids[t]];  // re-adding an id replaces it (reload_tracks, core/mod.rs:73-82)
        for (Channel &ch : tr.ch) free_channel(ch);
        tr.sr = srs[t];
        tr.ch.assign(n_channels[t], Channel());
        for (uint32_t k = 0; k < n_channels[t]; k++, flat++) {
            TH_REQUIRE(channels_flat[flat], "track %zu channel %u: NULL data", ids[t], k);
            Channel &ch = tr.ch[k];
            ch.n = n_samples[t];
            TH_HIP(hipMalloc((void **)&ch.d_wav, ch.n * sizeof(float)));
            TH_HIP(hipMemcpyAsync(ch.d_wav, channels_flat[flat], ch.n * sizeof(float), hipMemcpyHostToDevice, c->stream));
        }
        added.push_back(ids[t]);
    }
    TH_HIP(hipStreamSynchronize(c->stream));  // inputs are borrowed for this call only
    int rc = update_specs(tm, added);
    if (rc != TH_OK) return rc;
    tm->no_spec_img_ids.insert(tm->no_spec_img_ids.end(), added.begin(), added.end());  // core/mod.rs:70
    tm->invalidate_all();                                                                // lib.rs:192
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_add_track(th_tm *tm, size_t id, uint32_t sr, uint32_t n_channels, const float *const *channels,
                           size_t n_samples) {
    return th_tm_add_tracks(tm, 1, &id, &sr, &n_channels, channels, &n_samples);
}

TH_API int th_tm_remove_track(th_tm *tm, size_t id) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    auto it = tm->tracks.find(id);
    if (it == tm->tracks.end()) return fail(TH_ERR_NOT_FOUND, "Track %zu does not exist", id);
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    for (Channel &ch : it->second.ch) free_channel(ch);
    tm->tracks.erase(it);
    retain_plans(tm);      // core/mod.rs:96-99
    tm->invalidate_all();  // lib.rs:221
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_apply_track_list_changes(th_tm *tm, size_t *updated_ids, size_t cap, size_t *n_updated,
                                          uint32_t *max_sr) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    std::vector<size_t> upd;
    int rc = update_spec_imgs(tm, false, &upd);
    if (rc != TH_OK) return rc;
    if (n_updated) *n_updated = upd.size();
    if (updated_ids)
        for (size_t i = 0; i < upd.size() && i < cap; i++) updated_ids[i] = upd[i];
    if (max_sr) *max_sr = tm->max_sr;
    if (!upd.empty()) tm->invalidate_spectrogram();  // lib.rs:243-245
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_get_db_state(const th_tm *tm, float *min_dB, float *max_dB, uint32_t *max_sr) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    if (min_dB) *min_dB = tm->min_dB;
    if (max_dB) *max_dB = tm->max_dB;
    if (max_sr) *max_sr = tm->max_sr;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_spec_shape(const th_tm *tm, size_t id, uint32_t ch, size_t *n_frames, size_t *height) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    const Channel *c = find_channel(const_cast<th_tm *>(tm), id, ch);
    if (!c || !c->has_spec) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    if (n_frames) *n_frames = c->T;
    if (height) *height = c->H;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_img_shape(const th_tm *tm, size_t id, uint32_t ch, size_t *img_height, size_t *img_width) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    const Channel *c = find_channel(const_cast<th_tm *>(tm), id, ch);
    if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    if (img_height) *img_height = c->img_h;
    if (img_width) *img_width = c->img_w;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_copy_spec(th_tm *tm, size_t id, uint32_t ch, float *out, size_t cap) {
    TH_TRY
    TH_REQUIRE(tm && out, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    Channel *c = find_channel(tm, id, ch);
    if (!c || !c->has_spec) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    if (cap < c->T * c->H) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu floats", c->T * c->H);
    // de-pad to the reference's dense T x H layout
    if (c->T && c->H)
        TH_HIP(hipMemcpy2DAsync(out, c->H * sizeof(float), c->d_spec, c->spec_pitch * sizeof(float), c->H * sizeof(float),
                                c->T, hipMemcpyDeviceToHost, tm->ctx->stream));
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_copy_img(th_tm *tm, size_t id, uint32_t ch, uint16_t *out, size_t cap) {
    TH_TRY
    TH_REQUIRE(tm && out, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    TH_HIP(hipSetDevice(tm->ctx->device));
    Channel *c = find_channel(tm, id, ch);
    if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    if (cap < c->img_h * c->img_w) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu pixels", c->img_h * c->img_w);
    if (c->img_h && c->img_w)
        TH_HIP(hipMemcpy2DAsync(out, c->img_w * sizeof(uint16_t), c->d_img, c->img_pitch * sizeof(uint16_t),
                                c->img_w * sizeof(uint16_t), c->img_h, hipMemcpyDeviceToHost, tm->ctx->stream));
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_revisions(const th_tm *tm, uint64_t *waveform_revision, uint64_t *spectrogram_revision) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    if (waveform_revision) *waveform_revision = tm->waveform_revision();
    if (spectrogram_revision) *spectrogram_revision = tm->spectrogram_revision();
    return TH_OK;
    TH_CATCH
}

// get_spectrogram_tile — lib.rs:369-389 → RenderTileCache::spectrogram_tile (render_tiles.rs:171-188)
TH_API int th_tm_get_spectrogram_tile(th_tm *tm, size_t id, uint32_t ch, uint32_t level_x, uint32_t level_y,
                                      uint32_t tile_x, uint32_t tile_y, uint8_t *out, size_t cap, size_t *out_len) {
    TH_TRY
    TH_REQUIRE(tm && out && out_len, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    Channel *c = find_channel(tm, id, ch);
    if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    return th_encode_spectrogram_tile_dev(tm->ctx, c->d_img, c->img_h, c->img_w, c->img_pitch, tm->colormap_rgba.data(),
                                          tm->colormap_rgba.size(), tm->spectrogram_revision(), level_x, level_y,
                                          tile_x, tile_y, out, cap, out_len);
    TH_CATCH
}

// get_waveform_tile — lib.rs:342-367: cache lookup (:350-355), encode on a miss, store (:358-365)
TH_API int th_tm_get_waveform_tile(th_tm *tm, size_t id, uint32_t ch, uint32_t level, uint32_t tile_index,
                                   uint8_t *out, size_t cap, size_t *out_len) {
    TH_TRY
    TH_REQUIRE(tm && out && out_len, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    Channel *c = find_channel(tm, id, ch);
    if (!c) return fail(TH_ERR_NOT_FOUND, "Track %zu does not exist", id);
    uint64_t revision = 0;
    std::vector<uint8_t> cached;
    if (tm->cache.lookup(id, ch, level, tile_index, &revision, &cached)) {
        *out_len = cached.size();
        if (cap < cached.size()) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", cached.size());
        std::memcpy(out, cached.data(), cached.size());
        return TH_OK;
    }
    const int rc = th_encode_waveform_tile_dev(tm->ctx, c->d_wav, c->n, revision, level, tile_index, out, cap, out_len);
    if (rc == TH_OK) tm->cache.store(id, ch, revision, level, tile_index, out, *out_len);
    return rc;
    TH_CATCH
}

// get_audio_render_metadata — lib.rs:321-340 → RenderTileCache::metadata (render_tiles.rs:101-122)
TH_API int th_tm_get_audio_render_metadata(th_tm *tm, size_t id, uint32_t ch, double track_sec, int is_clipped,
                                           th_render_metadata *out) {
    TH_TRY
    TH_REQUIRE(tm && out, "NULL argument");
    std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);
    auto it = tm->tracks.find(id);
    if (it == tm->tracks.end() || ch >= it->second.ch.size()) return fail(TH_ERR_NOT_FOUND, "Track %zu does not exist", id);
    const Channel &c = it->second.ch[ch];
    *out = th_render_metadata{};
    out->waveform_revision = tm->waveform_revision();
    out->spectrogram_revision = tm->spectrogram_revision();
    out->sample_rate = it->second.sr;
    out->is_clipped = is_clipped ? 1u : 0u;
    out->sample_count = c.n;
    out->track_sec = track_sec;
    out->spectrogram_height = c.d_img ? c.img_h : 0;  // (shape[0], shape[1]) of the u16 image, lib.rs:330-332
    out->spectrogram_width = c.d_img ? c.img_w : 0;
    out->waveform_tile_bins = 1024;
    out->spectrogram_tile_size = 512;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_tile_cache(th_tm *tm, th_tile_cache **out) {
    TH_TRY
    TH_REQUIRE(tm && out, "NULL argument");
    *out = &tm->cache;
    return TH_OK;
    TH_CATCH
}
