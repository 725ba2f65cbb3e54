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
#include <condition_variable>
#include <cstring>
#include <chrono>
#include <list>
#include <map>
#include <memory>
#include <set>
#include <thread>
#include <atomic>
#include <shared_mutex>
#include <tuple>
#include <vector>

#include "common.h"
#include "context.h"
#include "host_math.h"
#include "kernels.h"
#include "tile_cache.h"

using namespace th;

namespace {

// One level of a channel's LOD mip pyramid: the whole u16 image resized to ceil(W / 2^lx) x ceil(H / 2^ly) with the
// separable Lanczos3 of resize_spectrogram_tile (render_tiles.rs:354-393).  A LOD tile is then a crop of it.
struct MipLevel {
    uint16_t *d = nullptr;
    uint32_t w = 0, h = 0, pitch = 0;
};

struct Channel {
    float *d_wav = nullptr;
    size_t n = 0;
    float *d_pyr = nullptr;      // resident waveform pyramid: every level's (min, max, mean) bins, th_waveform_pyramid_offset
    uint32_t pyr_levels = 0;     // levels 0 .. pyr_levels - 1 (the last one has a single bin)
    float *d_spec = nullptr;
    size_t T = 0, H = 0, spec_pitch = 0;   // rows padded to 128 B (th_pitch_f32)
    uint16_t *d_img = nullptr;
    size_t img_h = 0, img_w = 0, img_pitch = 0;  // rows padded to 128 B (th_pitch_u16)
    std::map<std::pair<uint32_t, uint32_t>, MipLevel> mips;  // (level_x, level_y) != (0, 0): views into d_mips
    uint16_t *d_mips = nullptr;                               // one allocation for every level
    size_t mips_elems = 0;                                    // its size (re-made images of the same shape reuse it)
    float mn = INFINITY, mx = -INFINITY;  // find_min_max of this spec (simd.rs:14-36)
    bool has_spec = false;
};

// Tile requests (lib.rs:342-389) are served concurrently, as the reference's IPC readers are (RwLock::read, lib.rs:345,378;
// thread-local Resizer, render_tiles.rs:366-368): each request borrows a reader slot — its own HIP stream, a device tile
// buffer and a pinned host staging buffer — so that no request waits for another's kernel or copy.
struct ReaderSlot {
    hipStream_t stream = nullptr;
    uint8_t *h_tile = nullptr;  // pinned, mapped into the device's address space
    uint8_t *h_tile_dev = nullptr;  // its device-side address
    bool busy = false;
};
constexpr size_t TILE_BYTES_MAX = 520 * 520 * 4;  // 512 core + 2 x 4 gutter (render_tiles.rs:15-16); >= 1024 * 12 waveform bins
constexpr size_t MAX_READER_SLOTS = 16;

// Lanczos tap table of one image axis at one LOD level, resident on the device (shared by every channel of that size)
// a device allocation shared by the tap tables that were made together (freed when the last of them goes)
struct DevBlock {
    void *p = nullptr;
    ~DevBlock() {
        if (p) (void)hipFree(p);
    }
};
struct AxisTable {
    std::shared_ptr<DevBlock> owner;  // d_blob points into it
    void *d_blob = nullptr;
    uint32_t n_out = 0, max_taps = 0;
    uint32_t span[3] = {0, 0, 0};  // LodAxis::span
    size_t bytes = 0;
};

struct Track {
    uint32_t sr = 0;
    std::vector<Channel> ch;
    // ONE device allocation for the samples and waveform pyramids of all channels (round 6: two hipMalloc per channel were 0.7 ms
    // of a 32-track add); Channel::d_wav / d_pyr are views into it and live exactly as long as the track
    void *d_pool = nullptr;
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
    uint8_t *d_colormap = nullptr;  // device copy of colormap_rgba (uploaded when it changes, not per tile request)
    th_tile_cache cache{th_tile_cache::DEFAULT_BUDGET};  // revisions + waveform-tile LRU, render_tiles.rs:51-230
    // Writers (every mutator) take `rw` exclusively, tile readers share it — the reference's RwLock<TrackManager>.
    // A mutator leaves the context stream idle before it releases the lock, so readers never see half-made images.
    mutable std::shared_mutex rw;
    std::mutex slot_mu;
    std::condition_variable slot_cv;
    std::vector<std::unique_ptr<ReaderSlot>> slots;
    // 0: LOD > 0 tiles are crops of the pre-built mip pyramid (default); 1: resampled per request from the level-0 image
    // (the reference's own flow; kept as the comparison path).  Levels the pyramid does not hold always use 1.
    int lod_source = 0;
    std::map<std::pair<uint32_t, uint32_t>, AxisTable> axis_tabs;  // (source length, level) -> taps
    th::DeviceTable mip_jobs;                                       // job table of the batched mip-pyramid passes
    th::DeviceTable mip_scratch;                                    // transposed images of one chunk of the batch
    std::list<std::vector<unsigned char>> axis_staging;             // host copies of tap tables whose upload may still be in flight (cleared by writer_done)
    hipStream_t copy_stream = nullptr;                              // th_tm_add_tracks: uploads of group g + 1 beside the kernels of group g
    hipEvent_t copy_ev = nullptr;
    // update_spec_imgs records img_ev behind the quantiser and mips_ev behind the last pass of the mip pyramids.  The two
    // interactive mutators (apply_track_list_changes, set_dB_range) return when the IMAGES are made and leave the pyramids — an
    // acceleration structure of this library for LOD > 0 tiles, 2.7 of the 4 ms of a 32-track update; the reference has none —
    // building on the context stream: a tile reader that crops a mip level makes its own stream wait for mips_ev (on the
    // device, not on the host), level-0 tiles do not wait at all, and everything else that touches images or levels runs on
    // the context stream, i.e. behind the build anyway.
    hipEvent_t img_ev = nullptr, mips_ev = nullptr;
    bool img_ev_recorded = false;
    uint64_t stat_tiles_served = 0;
    // th_tm_get_spectrogram_tiles: pinned staging for callers whose buffer the GPU cannot write directly (one batch at a time)
    std::mutex batch_mu;
    void *batch_stage = nullptr, *batch_stage_dev = nullptr;
    size_t batch_stage_cap = 0;

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

void free_mips(Channel &c) {
    if (c.d_mips) (void)hipFree(c.d_mips);
    c.d_mips = nullptr;
    c.mips_elems = 0;
    c.mips.clear();
}

void free_channel(Channel &c) {  // (d_wav / d_pyr belong to the track's pool: free_track)
    if (c.d_spec) (void)hipFree(c.d_spec);
    if (c.d_img) (void)hipFree(c.d_img);
    free_mips(c);
    c = Channel();
}

void free_track(Track &t) {
    for (Channel &ch : t.ch) free_channel(ch);
    t.ch.clear();
    if (t.d_pool) (void)hipFree(t.d_pool);
    t.d_pool = nullptr;
}

struct Setting {
    double win_ms;
    uint32_t t_overlap, f_overlap;
    int freq_scale;
};
Setting setting_of(const th_tm *tm) { return Setting{tm->win_ms, tm->t_overlap, tm->f_overlap, tm->freq_scale}; }

PlanKey plan_key(const Setting &st, uint32_t sr) {
    size_t hop, win, n_fft;
    calc_framing_params(st.win_ms, st.t_overlap, st.f_overlap, sr, &hop, &win, &n_fft);
    return PlanKey{sr, win, hop, n_fft, st.freq_scale};
}

// the plan of (setting, sr): the cached one, or a new one recorded in `created` (the caller commits or destroys those)
int get_plan(th_tm *tm, const Setting &st, uint32_t sr, std::map<PlanKey, th_plan *> &created, th_plan **out) {
    const PlanKey key = plan_key(st, sr);
    auto it = tm->plans.find(key);
    if (it != tm->plans.end()) {
        *out = it->second;
        return TH_OK;
    }
    auto ic = created.find(key);
    if (ic != created.end()) {
        *out = ic->second;
        return TH_OK;
    }
    th_plan *p = nullptr;
    int rc = th_plan_create(tm->ctx, sr, std::get<1>(key), std::get<2>(key), std::get<3>(key), st.freq_scale, 0, &p);
    if (rc != TH_OK) return rc;
    created[key] = p;
    *out = p;
    return TH_OK;
}

// SpectrogramAnalyzer::retain — spectrogram.rs:156-185: drop plans no resident track needs
void retain_plans(th_tm *tm) {
    std::set<PlanKey> need;
    const Setting st = setting_of(tm);
    for (auto &kv : tm->tracks) need.insert(plan_key(st, kv.second.sr));
    for (auto it = tm->plans.begin(); it != tm->plans.end();) {
        if (!need.count(it->first)) {
            th_plan_destroy(it->second);
            it = tm->plans.erase(it);
        } else {
            ++it;
        }
    }
}

// A freshly computed spec of one channel, not yet attached to it
struct NewSpec {
    Channel *ch = nullptr;
    float *d_spec = nullptr;
    size_t T = 0, H = 0, pitch = 0;
    float mn = INFINITY, mx = -INFINITY;
};
void free_new_specs(std::vector<NewSpec> &v) {
    for (NewSpec &n : v)
        if (n.d_spec) (void)hipFree(n.d_spec);
    v.clear();
}

// TrackManager::update_specs — core/mod.rs:137-164: one batched launch per plan, INTO FRESH BUFFERS.  Nothing the
// manager owns is touched: the caller attaches the results (commit_specs) once every launch has succeeded, so a
// failure (unsupported n_fft, out of memory) leaves settings, specs and has_spec exactly as they were.
// pending != NULL: the per-channel (min, max) pairs stay on the device (entries appended to *pending) and NOTHING here waits for
// the launches — the caller synchronises once and calls finish_specs (th_tm_add_tracks launches group after group this way,
// beside the next group's upload)
struct PendingMinMax {
    float *d_mm = nullptr;
    size_t first = 0, count = 0;
};
int finish_specs(th_tm *tm, std::vector<PendingMinMax> &pending, std::vector<NewSpec> *out, bool read) {
    int rc = TH_OK;
    for (PendingMinMax &pm : pending) {
        if (read && rc == TH_OK) {
            std::vector<float> mm(2 * pm.count);
            const hipError_t e = hipMemcpy(mm.data(), pm.d_mm, mm.size() * sizeof(float), hipMemcpyDeviceToHost);
            if (e != hipSuccess) rc = th::fail(TH_ERR_HIP, "%s", hipGetErrorString(e));
            else
                for (size_t i = 0; i < pm.count; i++) {
                    (*out)[pm.first + i].mn = mm[2 * i];
                    (*out)[pm.first + i].mx = mm[2 * i + 1];
                }
        }
        (void)hipFree(pm.d_mm);
    }
    pending.clear();
    (void)tm;
    return rc;
}
int compute_specs(th_tm *tm, const Setting &st, const std::vector<std::pair<uint32_t, Channel *>> &chans,
                  std::map<PlanKey, th_plan *> &created, std::vector<NewSpec> *out, std::vector<PendingMinMax> *pending = nullptr) {
    th_ctx *c = tm->ctx;
    std::map<th_plan *, std::vector<Channel *>> groups;
    for (auto &sc : chans) {
        th_plan *p = nullptr;
        int rc = get_plan(tm, st, sc.first, created, &p);
        if (rc != TH_OK) return rc;
        groups[p].push_back(sc.second);
    }
    for (auto &kv : groups) {
        th_plan *p = kv.first;
        std::vector<Channel *> &chs = kv.second;
        std::vector<th_chan_desc> descs(chs.size());
        const size_t first = out->size();
        for (size_t i = 0; i < chs.size(); i++) {
            NewSpec ns;
            ns.ch = chs[i];
            ns.T = stft_n_frames(chs[i]->n, p->g.win, p->g.hop);
            ns.H = p->g.height;
            ns.pitch = th_pitch_f32(ns.H);
            out->push_back(ns);  // (recorded before the allocation: the caller frees whatever is in *out)
            TH_HIP(hipMalloc((void **)&out->back().d_spec, std::max<size_t>(1, ns.T * ns.pitch) * sizeof(float)));
            descs[i] = th_chan_desc{chs[i]->d_wav, out->back().d_spec, chs[i]->n, ns.T, ns.pitch};
        }
        float *d_mm = nullptr;
        TH_HIP(hipMalloc((void **)&d_mm, 2 * chs.size() * sizeof(float)));
        if (pending != nullptr) {
            pending->push_back(PendingMinMax{d_mm, first, chs.size()});  // (recorded before the launch: the caller frees it on any outcome)
            const int prc = th_calc_spec_batch_dev(p, descs.data(), descs.size(), d_mm);
            if (prc != TH_OK) return prc;
            continue;
        }
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
            (*out)[first + i].mn = mm[2 * i];
            (*out)[first + i].mx = mm[2 * i + 1];
        }
    }
    return TH_OK;
}

void commit_specs(std::vector<NewSpec> &v) {
    for (NewSpec &n : v) {
        Channel &ch = *n.ch;
        if (ch.d_spec) (void)hipFree(ch.d_spec);
        ch.d_spec = n.d_spec;
        ch.T = n.T;
        ch.H = n.H;
        ch.spec_pitch = n.pitch;
        ch.mn = n.mn;
        ch.mx = n.mx;
        ch.has_spec = true;
        n.d_spec = nullptr;
    }
    v.clear();
}

// ---------------------------------------------------------------------------------------------- LOD mip pyramid
// (levels beyond these are resampled per request: the frontend's level = floor(log2(source pixels per screen pixel)),
// AudioTrackViewport.tsx:91,406,439 — rows are hundreds of pixels high, so level_y rarely exceeds 2)
constexpr uint32_t MIP_MAX_LX = 12, MIP_MAX_LY = 3, MIP_MIN_DIM = 16;
constexpr uint32_t PYR_FIRST = 2;  // first materialised level of the resident waveform pyramids (levels 0, 1: from the samples)

// The host half of a tap table: Lanczos3 taps of the whole axis as the crop box (origin 0, extent n_in:
// resize_spectrogram_tile with the full image as the crop), packed as LodAxis reads it, + the span of R = 8, 4, 2 outputs
struct AxisHostTable {
    std::vector<unsigned char> blob;
    uint32_t n_out = 0, max_taps = 0, span[3] = {0, 0, 0};
};
void build_axis_host(uint32_t n_in, uint32_t level, AxisHostTable &h) {
    const size_t n_out = (n_in + ((size_t)1 << level) - 1) >> level;
    LodAxisHost ax;
    build_lod_axis(0.0, (double)n_in, n_out, 0, (long)n_in, ax);
    h.blob.resize(ax.blob_bytes(n_out));
    ax.pack(h.blob.data(), n_out);
    h.n_out = (uint32_t)n_out;
    h.max_taps = ax.max_taps;
    for (int k = 0; k < 3; k++) {  // source rows under the tap windows of R = 8, 4, 2 consecutive outputs
        const size_t r = (size_t)8 >> k;
        int64_t span = 0;
        for (size_t o0 = 0; o0 < n_out; o0 += r) {
            int64_t hi = 0;
            for (size_t o = o0; o < std::min(n_out, o0 + r); o++) hi = std::max<int64_t>(hi, (int64_t)ax.start[o] + ax.count[o]);
            span = std::max(span, hi - (int64_t)ax.start[o0]);
        }
        h.span[k] = (uint32_t)span;
    }
}

// Make every tap table in `keys` that is not resident yet: the host halves side by side on a few threads (a table is ~6 taps per
// SOURCE element whatever the level — 17 k sinc evaluations for a 30 s track's x axis — and a batch of tracks needs a dozen
// levels per distinct length: 2.1 of the 3.7 ms an apply_track_list_changes took for 32 equal tracks, and proportionally more
// for tracks of different lengths), then the uploads, stream-ordered in front of the passes that read them.  The blobs stay
// alive in tm->axis_staging until the writer's final synchronisation (writer_done): nothing waits per table.
int make_axis_tables(th_tm *tm, const std::set<std::pair<uint32_t, uint32_t>> &keys) {
    std::vector<std::pair<uint32_t, uint32_t>> todo;
    for (auto &k : keys)
        if (!tm->axis_tabs.count(k)) todo.push_back(k);
    if (todo.empty()) return TH_OK;
    std::vector<AxisHostTable> host(todo.size());
    {
        const size_t n_thr = std::min<size_t>(todo.size(), std::max(1u, std::min(16u, std::thread::hardware_concurrency())));
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&]() {
            try {
                for (size_t i = next.fetch_add(1); i < todo.size(); i = next.fetch_add(1)) build_axis_host(todo[i].first, todo[i].second, host[i]);
            } catch (...) {
                failed = true;
            }
        };
        std::vector<std::thread> thr;
        try {
            for (size_t i = 1; i < n_thr; i++) thr.emplace_back(work);
        } catch (...) {  // (no more threads: the ones that started and this one do all of it)
        }
        work();
        for (auto &t : thr) t.join();
        if (failed) return th::fail(TH_ERR_OOM, "host allocation failed while building LOD tap tables");
    }
    // ONE allocation and ONE upload for the tables of this call (fifteen small pageable copies were 0.3 ms of host time)
    std::vector<size_t> at(todo.size());
    size_t total = 0;
    for (size_t i = 0; i < todo.size(); i++) {
        at[i] = total;
        total += (host[i].blob.size() + 255) / 256 * 256;
    }
    tm->axis_staging.emplace_back(total);
    std::vector<unsigned char> &stage = tm->axis_staging.back();
    for (size_t i = 0; i < todo.size(); i++) std::memcpy(stage.data() + at[i], host[i].blob.data(), host[i].blob.size());
    std::shared_ptr<DevBlock> blk = std::make_shared<DevBlock>();
    TH_HIP(hipMalloc(&blk->p, total));
    const hipError_t e = hipMemcpyAsync(blk->p, stage.data(), total, hipMemcpyHostToDevice, tm->ctx->stream);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(tm->ctx->stream);
        TH_HIP(e);
    }
    for (size_t i = 0; i < todo.size(); i++) {
        AxisTable t;
        t.owner = blk;
        t.d_blob = static_cast<unsigned char *>(blk->p) + at[i];
        t.n_out = host[i].n_out;
        t.max_taps = host[i].max_taps;
        t.bytes = host[i].blob.size();
        for (int k = 0; k < 3; k++) t.span[k] = host[i].span[k];
        tm->axis_tabs.emplace(todo[i], t);
    }
    return TH_OK;
}

int axis_table(th_tm *tm, uint32_t n_in, uint32_t level, AxisTable **out) {
    const auto key = std::make_pair(n_in, level);
    auto it = tm->axis_tabs.find(key);
    if (it == tm->axis_tabs.end()) {
        const int rc = make_axis_tables(tm, {key});
        if (rc != TH_OK) return rc;
        it = tm->axis_tabs.find(key);
    }
    *out = &it->second;
    return TH_OK;
}

LodAxis axis_view(const AxisTable &t) {
    unsigned char *base = static_cast<unsigned char *>(t.d_blob);
    LodAxis a;
    a.start = reinterpret_cast<const int32_t *>(base);
    a.count = reinterpret_cast<const int32_t *>(base + (size_t)t.n_out * 4);
    a.wsum = reinterpret_cast<const double *>(base + (size_t)t.n_out * 8);
    a.w = reinterpret_cast<const double *>(base + (size_t)t.n_out * 16);
    a.n_out = t.n_out;
    a.max_taps = t.max_taps;
    for (int k = 0; k < 3; k++) a.span[k] = t.span[k];
    return a;
}

// Tap tables are keyed by (axis length, level); the x-axis length is a track's frame count, i.e. different for practically
// every track.  Drop the tables no resident image refers to (they survived remove_track / set_setting, ~48 W bytes
// per level each, so a long session that adds and removes tracks grew device memory without bound).  hipFree waits for the
// device, so a table still read by a launched pass is safe.
void prune_axis_tabs(th_tm *tm) {
    std::set<std::pair<uint32_t, uint32_t>> live;
    for (auto &kv : tm->tracks)
        for (Channel &ch : kv.second.ch)
            for (auto &m : ch.mips) {
                if (m.first.first) live.insert({(uint32_t)ch.img_w, m.first.first});
                if (m.first.second) live.insert({(uint32_t)ch.img_h, m.first.second});
            }
    for (auto it = tm->axis_tabs.begin(); it != tm->axis_tabs.end();) {
        if (live.count(it->first)) {
            ++it;
        } else {
            it = tm->axis_tabs.erase(it);  // (its block goes with the last table made beside it)
        }
    }
}

// Build every (lx, ly) level of the images of `chans`: horizontal pass from level 0 for each lx, vertical pass from
// (lx, 0) for each ly — the same order and the same one-rounding-per-pass as the per-request resize.  Images of one shape
// go through every pass together (grid z = image): a launch per level, not per level and channel (32 tracks: ~1000 launches
// of a few microseconds of work each took 27 ms of the 28 ms an apply_track_list_changes spent behind the STFT).
int build_mips(th_tm *tm, const std::vector<Channel *> &chans) {
    hipStream_t s = tm->ctx->stream;
    if (tm->lod_source == 1) {  // per-request route selected: no pyramid is kept (2.75x the image memory, and the passes)
        for (Channel *ch : chans) free_mips(*ch);
        return TH_OK;
    }
    struct Shape {
        uint32_t w, h, pitch;
        bool operator<(const Shape &o) const { return std::tie(w, h, pitch) < std::tie(o.w, o.h, o.pitch); }
    };
    std::map<Shape, std::vector<Channel *>> groups;
    for (Channel *ch : chans) {
        if (!ch->d_img || !ch->img_w || !ch->img_h) {
            free_mips(*ch);
            continue;
        }
        ch->mips.clear();
        const uint32_t W = (uint32_t)ch->img_w, Hh = (uint32_t)ch->img_h;
        uint32_t Lx = 0, Ly = 0;
        while (Lx < MIP_MAX_LX && ((W + (2u << Lx) - 1) >> (Lx + 1)) >= MIP_MIN_DIM) Lx++;
        while (Ly < MIP_MAX_LY && ((Hh + (2u << Ly) - 1) >> (Ly + 1)) >= MIP_MIN_DIM) Ly++;
        // shapes first, then ONE allocation for all levels (a hipMalloc per level cost more than the resampling)
        size_t total = 0;
        for (uint32_t lx = 0; lx <= Lx; lx++)
            for (uint32_t ly = 0; ly <= Ly; ly++) {
                if (!lx && !ly) continue;
                MipLevel m;
                m.w = (W + (1u << lx) - 1) >> lx;
                m.h = (Hh + (1u << ly) - 1) >> ly;
                m.pitch = (uint32_t)th_pitch_u16(m.w);
                m.d = reinterpret_cast<uint16_t *>(total);  // offset for now
                total += ((size_t)m.h * m.pitch + 127) / 128 * 128;
                ch->mips[{lx, ly}] = m;
            }
        if (!total) {
            free_mips(*ch);
            continue;
        }
        // (a dB-range or colour-map change re-makes every image at its old shape: keep the allocation — hipFree
        // synchronises the device and a hipMalloc of this size takes longer than the resampling)
        if (!ch->d_mips || ch->mips_elems != total) {
            const std::map<std::pair<uint32_t, uint32_t>, MipLevel> views = ch->mips;
            free_mips(*ch);
            hipError_t e = hipMalloc((void **)&ch->d_mips, total * sizeof(uint16_t));
            if (e != hipSuccess) TH_HIP(e);
            ch->mips = views;
            ch->mips_elems = total;
        }
        for (auto &kv : ch->mips) kv.second.d = ch->d_mips + reinterpret_cast<size_t>(kv.second.d);
        groups[Shape{W, Hh, (uint32_t)ch->img_pitch}].push_back(ch);
    }
    // One job table for all launches (uploaded once), then the launches in dependency order per shape.  The horizontal
    // pass runs as transpose -> vertical pass along x -> transpose back (kernels_image.hip: same taps, same order, same
    // rounding: bit-identical to lod_hpass_kernel); the transposed level-0 image and the transposed result of one level
    // live in a scratch buffer of the manager, MIP_CHUNK images at a time.
    enum Kind { TRANSPOSE, VPASS };
    struct Launch {
        Kind kind;
        uint32_t level, first, count, n_in, a, b;  // TRANSPOSE: a x b image;  VPASS: axis of length n_in at `level`, dw = a
    };
    constexpr size_t MIP_CHUNK = 32;
    std::vector<LodPassJob> jobs;
    std::vector<Launch> launches;
    size_t scratch_elems = 0;
    for (auto &g : groups) {
        const uint32_t W = g.first.w, Hh = g.first.h;
        uint32_t Lx = 0;
        for (auto &kv : g.second[0]->mips) Lx = std::max(Lx, kv.first.first);
        if (Lx == 0) continue;
        const size_t pT = th_pitch_u16(Hh);  // row pitch of the transposed images (rows = x, columns = y)
        const size_t per_img = (size_t)W * pT + (size_t)((W + 1) / 2) * pT;
        scratch_elems = std::max(scratch_elems, per_img * std::min(MIP_CHUNK, g.second.size()));
    }
    if (scratch_elems) {
        int rc = tm->mip_scratch.ensure(scratch_elems * sizeof(uint16_t));
        if (rc != TH_OK) return rc;
    }
    uint16_t *const scratch = static_cast<uint16_t *>(tm->mip_scratch.dptr);
    for (auto &g : groups) {
        const uint32_t W = g.first.w, Hh = g.first.h;
        uint32_t Lx = 0, Ly = 0;
        for (auto &kv : g.second[0]->mips) {
            Lx = std::max(Lx, kv.first.first);
            Ly = std::max(Ly, kv.first.second);
        }
        const uint32_t pT = (uint32_t)th_pitch_u16(Hh);
        const size_t t0_elems = (size_t)W * pT, per_img = t0_elems + (size_t)((W + 1) / 2) * pT;
        for (size_t c0 = 0; c0 < g.second.size(); c0 += MIP_CHUNK) {
            const std::vector<Channel *> cs(g.second.begin() + c0, g.second.begin() + std::min(g.second.size(), c0 + MIP_CHUNK));
            const uint32_t nc = (uint32_t)cs.size();
            auto t0 = [&](size_t i) { return scratch + i * per_img; };            // transposed level 0: W rows of Hh
            auto tl = [&](size_t i) { return scratch + i * per_img + t0_elems; };  // transposed level lx: w_lx rows of Hh
            if (Lx > 0) {
                launches.push_back(Launch{TRANSPOSE, 0, (uint32_t)jobs.size(), nc, 0, W, Hh});
                for (size_t i = 0; i < nc; i++) jobs.push_back(LodPassJob{cs[i]->d_img, t0(i), (uint32_t)cs[i]->img_pitch, pT});
            }
            for (uint32_t lx = 0; lx <= Lx; lx++) {
                if (lx > 0) {
                    const uint32_t wl = cs[0]->mips[{lx, 0u}].w;
                    launches.push_back(Launch{VPASS, lx, (uint32_t)jobs.size(), nc, W, Hh, 0});  // along x, Hh columns
                    for (size_t i = 0; i < nc; i++) jobs.push_back(LodPassJob{t0(i), tl(i), pT, pT});
                    launches.push_back(Launch{TRANSPOSE, 0, (uint32_t)jobs.size(), nc, 0, Hh, wl});  // (Hh x wl)^T -> wl x Hh image
                    for (size_t i = 0; i < nc; i++) {
                        const MipLevel &m = cs[i]->mips[{lx, 0u}];
                        jobs.push_back(LodPassJob{tl(i), m.d, pT, m.pitch});
                    }
                }
                for (uint32_t ly = 1; ly <= Ly; ly++) {
                    const uint32_t src_w = lx > 0 ? cs[0]->mips[{lx, 0u}].w : W;
                    launches.push_back(Launch{VPASS, ly, (uint32_t)jobs.size(), nc, Hh, src_w, 0});
                    for (Channel *ch : cs) {
                        const MipLevel &m = ch->mips[{lx, ly}];
                        if (lx > 0) {
                            const MipLevel &src = ch->mips[{lx, 0u}];
                            jobs.push_back(LodPassJob{src.d, m.d, src.pitch, m.pitch});
                        } else {
                            jobs.push_back(LodPassJob{ch->d_img, m.d, (uint32_t)ch->img_pitch, m.pitch});
                        }
                    }
                }
            }
        }
    }
    if (launches.empty()) return TH_OK;
    const bool prof = getenv("TH_TM_PROF") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tb0 = now();
    double t_axis = 0.0;
    int rc = tm->mip_jobs.upload(s, jobs.data(), jobs.size() * sizeof(LodPassJob));
    if (rc != TH_OK) return rc;
    const LodPassJob *dj = static_cast<const LodPassJob *>(tm->mip_jobs.dptr);
    {   // every tap table the passes below read, made together (in parallel on the host) before the first launch
        std::set<std::pair<uint32_t, uint32_t>> keys;
        for (const Launch &l : launches)
            if (l.kind == VPASS) keys.insert({l.n_in, l.level});
        const double ta_ = now();
        rc = make_axis_tables(tm, keys);
        t_axis += now() - ta_;
        if (rc != TH_OK) return rc;
    }
    for (const Launch &l : launches) {
        if (l.kind == TRANSPOSE) {
            // a x b: the SOURCE has b rows of a elements (dst[x][y] = src[y][x], x < a, y < b)
            TH_HIP(launch_transpose_u16_batch(dj + l.first, l.count, l.a, l.b, s));
        } else {
            AxisTable *t = nullptr;
            const double ta_ = now();
            rc = axis_table(tm, l.n_in, l.level, &t);
            t_axis += now() - ta_;
            if (rc != TH_OK) return rc;
            TH_HIP(launch_lod_vpass_batch(dj + l.first, l.count, axis_view(*t), l.a, s));
        }
    }
    if (prof) fprintf(stderr, "build_mips prof: %zu launches %.2f ms of which axis tables %.2f\n", launches.size(), now() - tb0, t_axis);
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
                    free_mips(ch);
                }
    std::vector<th_img_desc> descs;
    std::vector<Channel *> made;
    const bool prof = getenv("TH_TM_PROF") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tq0 = now();
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
            made.push_back(&ch);
        }
    }
    const double tq1 = now();
    int rc = th_spec_to_img_batch_dev(tm->ctx, descs.data(), descs.size(), tm->min_dB, tm->max_dB, tm->colormap_length);
    if (rc == TH_OK) {
        TH_HIP(hipEventRecord(tm->img_ev, tm->ctx->stream));
        tm->img_ev_recorded = true;
    }
    const double tq2 = now();
    // the LOD mip pyramid of every image that was just re-made (render_tiles.rs:290-313,354-393)
    if (rc == TH_OK) rc = build_mips(tm, made);
    if (rc == TH_OK) TH_HIP(hipEventRecord(tm->mips_ev, tm->ctx->stream));
    if (prof) {
        const double tq3 = now();
        (void)hipStreamSynchronize(tm->ctx->stream);
        fprintf(stderr, "update_spec_imgs prof: image allocations %.2f ms, quantise launch %.2f, build_mips (host side) %.2f, drain %.2f\n", tq1 - tq0, tq2 - tq1,
                tq3 - tq2, now() - tq3);
    }
    if (rc != TH_OK) {
        // Not failure-atomic otherwise: the images above are already re-quantised, so a pyramid that was only
        // partly rebuilt would serve pixels of the old dB range under the new revision.  Without levels every LOD request
        // falls back to the per-request resize of the (new) level-0 image.
        (void)hipStreamSynchronize(tm->ctx->stream);
        for (Channel *ch : made) free_mips(*ch);
    }
    prune_axis_tabs(tm);
    return rc;
}

Channel *find_channel(th_tm *tm, size_t id, uint32_t ch) {
    auto it = tm->tracks.find(id);
    if (it == tm->tracks.end() || ch >= it->second.ch.size()) return nullptr;
    return &it->second.ch[ch];
}

// ---------------------------------------------------------------------------------------------- reader slots
struct SlotLease {
    th_tm *tm;
    ReaderSlot *slot;
    ~SlotLease() {
        if (slot) {
            {
                std::lock_guard<std::mutex> lk(tm->slot_mu);
                slot->busy = false;
            }
            tm->slot_cv.notify_one();
        }
    }
};

int acquire_slot(th_tm *tm, ReaderSlot **out) {
    std::unique_lock<std::mutex> lk(tm->slot_mu);
    for (;;) {
        for (auto &sp : tm->slots)
            if (!sp->busy) {
                sp->busy = true;
                *out = sp.get();
                return TH_OK;
            }
        if (tm->slots.size() < MAX_READER_SLOTS) {
            std::unique_ptr<ReaderSlot> sl(new ReaderSlot());
            hipError_t e = hipStreamCreateWithFlags(&sl->stream, hipStreamNonBlocking);
            if (e == hipSuccess) e = hipHostMalloc((void **)&sl->h_tile, TILE_BYTES_MAX, hipHostMallocMapped);
            if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&sl->h_tile_dev, sl->h_tile, 0);
            if (e != hipSuccess) {
                if (sl->h_tile) (void)hipHostFree(sl->h_tile);
                if (sl->stream) (void)hipStreamDestroy(sl->stream);
                if (tm->slots.empty()) TH_HIP(e);
                // out of resources with slots in use: wait for one of those instead
            } else {
                sl->busy = true;
                *out = sl.get();
                tm->slots.push_back(std::move(sl));
                return TH_OK;
            }
        }
        tm->slot_cv.wait(lk);
    }
}

void put_u32(uint8_t *p, uint32_t v) { std::memcpy(p, &v, 4); }  // little-endian host (x86-64)
void put_u64(uint8_t *p, uint64_t v) { std::memcpy(p, &v, 8); }

// a mutator's last act before it releases the write lock: readers use their own streams
int writer_done(th_tm *tm, int rc) {
    hipError_t e = hipStreamSynchronize(tm->ctx->stream);
    tm->axis_staging.clear();  // (every upload from these host copies has completed)
    tm->img_ev_recorded = false;
    if (rc == TH_OK && e != hipSuccess) TH_HIP(e);
    return rc;
}

// the same for the mutators that only re-make images (see th_tm::img_ev): wait for the quantiser, not for the mip pyramids behind
// it.  The tap tables' host copies (axis_staging) stay until a later mutator has drained the stream.
int writer_done_images(th_tm *tm, int rc) {
    if (rc != TH_OK || !tm->img_ev_recorded || tm->lod_source != 0) return writer_done(tm, rc);
    const hipError_t e = hipEventSynchronize(tm->img_ev);
    tm->img_ev_recorded = false;
    if (e != hipSuccess) TH_HIP(e);
    return TH_OK;
}

}  // namespace

TH_API int th_tm_create(th_ctx *c, th_tm **out) {
    TH_TRY
    TH_REQUIRE(c && out, "NULL argument");
    std::unique_ptr<th_tm> tm(new th_tm());
    tm->ctx = c;
    // the manager's own device objects up front (a stream takes milliseconds to create: not inside the first add_tracks)
    TH_HIP(hipSetDevice(c->device));
    hipError_t e = hipStreamCreateWithFlags(&tm->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&tm->copy_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&tm->img_ev, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&tm->mips_ev, hipEventDisableTiming);
    if (e != hipSuccess) {
        if (tm->mips_ev) (void)hipEventDestroy(tm->mips_ev);
        if (tm->img_ev) (void)hipEventDestroy(tm->img_ev);
        if (tm->copy_ev) (void)hipEventDestroy(tm->copy_ev);
        if (tm->copy_stream) (void)hipStreamDestroy(tm->copy_stream);
        TH_HIP(e);
    }
    *out = tm.release();
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_destroy(th_tm *tm) {
    TH_TRY
    if (!tm) return TH_OK;
    (void)hipSetDevice(tm->ctx->device);
    (void)hipStreamSynchronize(tm->ctx->stream);
    for (auto &sp : tm->slots) {
        (void)hipStreamSynchronize(sp->stream);
        (void)hipStreamDestroy(sp->stream);
        (void)hipHostFree(sp->h_tile);
    }
    for (auto &kv : tm->tracks)
        free_track(kv.second);
    for (auto &kv : tm->plans) th_plan_destroy(kv.second);
    tm->axis_tabs.clear();
    if (tm->batch_stage) (void)hipHostFree(tm->batch_stage);
    tm->mip_jobs.release();
    tm->mip_scratch.release();
    if (tm->d_colormap) (void)hipFree(tm->d_colormap);
    if (tm->copy_ev) (void)hipEventDestroy(tm->copy_ev);
    if (tm->img_ev) (void)hipEventDestroy(tm->img_ev);
    if (tm->mips_ev) (void)hipEventDestroy(tm->mips_ev);
    if (tm->copy_stream) {
        (void)hipStreamSynchronize(tm->copy_stream);
        (void)hipStreamDestroy(tm->copy_stream);
    }
    delete tm;
    return TH_OK;
    TH_CATCH
}

namespace {
int upload_colormap(th_tm *tm) {
    if (tm->d_colormap) {
        TH_HIP(hipFree(tm->d_colormap));
        tm->d_colormap = nullptr;
    }
    TH_HIP(hipMalloc((void **)&tm->d_colormap, tm->colormap_rgba.size()));
    TH_HIP(hipMemcpyAsync(tm->d_colormap, tm->colormap_rgba.data(), tm->colormap_rgba.size(), hipMemcpyHostToDevice,
                          tm->ctx->stream));
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    return TH_OK;
}
}  // namespace

TH_API int th_tm_set_colormap(th_tm *tm, const uint8_t *rgba, size_t bytes) {
    TH_TRY
    TH_REQUIRE(tm && rgba, "NULL argument");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    // RenderTileCache::set_colormap keeps the old map for malformed input — render_tiles.rs:80-85
    if (bytes >= 4 && bytes % 4 == 0) {
        tm->colormap_rgba.assign(rgba, rgba + bytes);
        int rc = upload_colormap(tm);
        if (rc != TH_OK) return rc;
    }
    tm->invalidate_spectrogram();
    // init: TM.set_colormap_length(len / 4) → update_spec_imgs(force) — lib.rs:61, core/mod.rs:128-131
    tm->colormap_length = (uint32_t)(bytes / 4);
    std::vector<size_t> upd;
    return writer_done(tm, update_spec_imgs(tm, true, &upd));
    TH_CATCH
}

TH_API int th_tm_set_setting(th_tm *tm, double win_ms, uint32_t t_overlap, uint32_t f_overlap, int freq_scale) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    TH_REQUIRE(win_ms > 0. && t_overlap >= 1 && f_overlap >= 1, "invalid SpecSetting (lib.rs:275-277)");
    TH_REQUIRE(freq_scale == TH_FREQ_LINEAR || freq_scale == TH_FREQ_MEL, "bad freq_scale");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    // Transactional: plans and specs of the NEW setting are made first, into fresh buffers; the manager's own state
    // changes only when all of it exists.  (The reference cannot fail here: realfft takes any even length.  This
    // library plans n_fft = 2^a * odd with a >= 1 and odd <= 63, up to TH_MAX_N_FFT (th_plan_create): f_overlap = 3, 5, 6, 7 run
    // on the generic kernel; e.g. f_overlap = 67 or an n_fft above 2^20 is refused and leaves everything as it was.)
    const Setting st{win_ms, t_overlap, f_overlap, freq_scale};
    std::vector<std::pair<uint32_t, Channel *>> chans;
    for (auto &kv : tm->tracks)
        for (Channel &ch : kv.second.ch) chans.push_back({kv.second.sr, &ch});
    std::map<PlanKey, th_plan *> created;
    std::vector<NewSpec> fresh;
    int rc = compute_specs(tm, st, chans, created, &fresh);
    if (rc != TH_OK) {
        (void)hipStreamSynchronize(tm->ctx->stream);
        free_new_specs(fresh);
        for (auto &kv : created) th_plan_destroy(kv.second);
        return rc;
    }
    tm->win_ms = win_ms;
    tm->t_overlap = t_overlap;
    tm->f_overlap = f_overlap;
    tm->freq_scale = freq_scale;
    for (auto &kv : created) tm->plans[kv.first] = kv.second;
    commit_specs(fresh);
    retain_plans(tm);  // spec_analyzer.retain(...)  core/mod.rs:111-112
    std::vector<size_t> upd;
    rc = update_spec_imgs(tm, true, &upd);
    tm->invalidate_spectrogram();  // lib.rs:284
    return writer_done(tm, rc);
    TH_CATCH
}

TH_API int th_tm_set_dB_range(th_tm *tm, float dB_range) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    TH_REQUIRE(dB_range > 0.f, "dB_range must be > 0 (lib.rs:259)");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    tm->dB_range = dB_range;
    std::vector<size_t> upd;
    int rc = update_spec_imgs(tm, true, &upd);
    tm->invalidate_spectrogram();  // lib.rs:265
    return writer_done_images(tm, rc);
    TH_CATCH
}

TH_API int th_tm_add_tracks(th_tm *tm, size_t n_tracks, const size_t *ids, const uint32_t *srs,
                            const uint32_t *n_channels, const float *const *channels_flat, const size_t *n_samples) {
    TH_TRY
    TH_REQUIRE(tm && ids && srs && n_channels && channels_flat && n_samples, "NULL argument");
    TH_REQUIRE(n_tracks >= 1, "no tracks (lib.rs:182)");
    {   // validate everything before anything is allocated
        size_t flat = 0;
        for (size_t t = 0; t < n_tracks; t++) {
            TH_REQUIRE(srs[t] > 0 && n_channels[t] >= 1 && n_samples[t] >= 1, "track %zu: empty or invalid", ids[t]);
            for (uint32_t k = 0; k < n_channels[t]; k++, flat++)
                TH_REQUIRE(channels_flat[flat], "track %zu channel %u: NULL data", ids[t], k);
        }
    }
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    th_ctx *c = tm->ctx;
    TH_HIP(hipSetDevice(c->device));
    // Transactional: the new tracks are staged (audio upload, waveform pyramid, spec) beside the resident ones and
    // swapped in when everything has succeeded.  A failure frees the staging area and changes nothing.
    // Pipelined (round 6; the reference overlaps decode and calc_spec across rayon tasks, core/track.rs:211-239 -> core/mod.rs:
    // 153-163): the tracks go up in GROUPS on a copy stream of the manager's own; the waveform pyramids and the STFT of group g run
    // on the context stream (behind an event of the copy stream) while the host feeds group g + 1 to the copy engine — a pageable
    // source keeps the calling thread inside hipMemcpyAsync for the length of the transfer, so the kernels of the group before
    // cost nothing; nothing waits for the device until the end (one synchronisation, then the per-channel (min, max) read-back).
    // Inputs stay borrowed until the call returns.
    std::map<size_t, Track> staged;
    std::map<PlanKey, th_plan *> created;
    std::vector<NewSpec> fresh;
    std::vector<PendingMinMax> pending;
    TH_REQUIRE(tm->copy_stream != nullptr && tm->copy_ev != nullptr, "manager without its copy stream");  // (th_tm_create)
    auto abort_staging = [&]() {
        (void)hipStreamSynchronize(tm->copy_stream);
        (void)hipStreamSynchronize(c->stream);
        (void)finish_specs(tm, pending, &fresh, false);
        free_new_specs(fresh);
        for (auto &kv : staged) free_track(kv.second);
        for (auto &kv : created) th_plan_destroy(kv.second);
    };
    int rc = TH_OK;
    hipError_t e = hipSuccess;
    std::vector<size_t> added;
    const bool prof = getenv("TH_TM_PROF") != nullptr;
    auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double tp0 = now();
    // the same id twice in one call: the later one wins, as sequential adds would — only its data is uploaded
    std::map<size_t, size_t> last_of;
    std::vector<size_t> flat0(n_tracks, 0);
    {
        size_t flat = 0;
        for (size_t t = 0; t < n_tracks; t++) {
            last_of[ids[t]] = t;
            flat0[t] = flat;
            flat += n_channels[t];
            added.push_back(ids[t]);
        }
    }
    // groups of about 96 MB of samples (at least one track each): a group's launches cost ~0.2 ms of host time (descriptor tables,
    // spec allocations: six groups of 32 MB were SLOWER than no pipeline, 6.0 against 5.1 ms for 32 tracks), the last group's
    // kernels are the only ones nothing hides.  (Round 6, tried: the uploads on a helper thread so that this thread's launches do
    // not hold up the copy engine — the launches then take 1.15 ms of host time instead of 0.67 and the call 5.2 instead of 4.9 ms:
    // the runtime serialises the pageable copy path and the launches of the two threads.)
    constexpr size_t GROUP_BYTES = (size_t)96 << 20;
    const Setting st = setting_of(tm);
    size_t t = 0;
    double t_launch = 0.0;
    while (t < n_tracks && e == hipSuccess && rc == TH_OK) {
        std::vector<std::pair<uint32_t, Channel *>> chans;
        std::vector<th_pyramid_desc> pdescs;
        size_t bytes = 0;
        for (; t < n_tracks && (bytes == 0 || bytes < GROUP_BYTES) && e == hipSuccess; t++) {
            if (last_of[ids[t]] != t) continue;
            Track &tr = staged[ids[t]];
            tr.sr = srs[t];
            tr.ch.assign(n_channels[t], Channel());
            // resident waveform pyramid: levels up to the one whose single bin spans the channel (render_tiles.rs:232-259)
            // (levels PYR_FIRST .. lv - 1: level 0 would be (x, x, x) per sample — half of the pyramid's bytes — and level 1 a
            // quarter; tiles of both are served from the resident samples instead, th_pyramid_desc.first_level)
            const size_t n = n_samples[t];
            uint32_t lv = 1;
            while (lv < PYR_MAX_LEVELS && ((uint64_t)1 << (lv - 1)) < n) lv++;
            const size_t wav_f = (n + 63) / 64 * 64;  // (256-byte pieces: every view starts on a 256-byte boundary)
            const size_t pyr_f = (std::max<size_t>(1, th_waveform_pyramid_offset(n, std::max(lv, PYR_FIRST)) - th_waveform_pyramid_offset(n, PYR_FIRST)) + 63) / 64 * 64;
            e = hipMalloc(&tr.d_pool, (wav_f + pyr_f) * n_channels[t] * sizeof(float));
            for (uint32_t k = 0; k < n_channels[t] && e == hipSuccess; k++) {
                Channel &ch = tr.ch[k];
                ch.n = n;
                ch.pyr_levels = lv;
                ch.d_wav = static_cast<float *>(tr.d_pool) + (size_t)k * (wav_f + pyr_f);
                ch.d_pyr = ch.d_wav + wav_f;
                e = hipMemcpyAsync(ch.d_wav, channels_flat[flat0[t] + k], ch.n * sizeof(float), hipMemcpyHostToDevice, tm->copy_stream);
                bytes += ch.n * sizeof(float);
                // (std::map nodes and this vector do not move any more: the pointer stays valid while later groups are staged)
                chans.push_back({tr.sr, &ch});
                pdescs.push_back(th_pyramid_desc{ch.d_wav, ch.d_pyr, ch.n, ch.pyr_levels, PYR_FIRST});
            }
        }
        if (e != hipSuccess) break;
        if (chans.empty()) continue;
        const double tl0 = now();
        e = hipEventRecord(tm->copy_ev, tm->copy_stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, tm->copy_ev, 0);
        if (e != hipSuccess) break;
        rc = th_waveform_pyramid_dev(c, pdescs.data(), pdescs.size());
        if (rc == TH_OK) rc = compute_specs(tm, st, chans, created, &fresh, &pending);
        t_launch += now() - tl0;
    }
    const double tp1 = now();
    if (e == hipSuccess && rc == TH_OK) e = hipStreamSynchronize(tm->copy_stream);  // inputs are borrowed for this call only
    if (e == hipSuccess && rc == TH_OK) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess || rc != TH_OK) {
        abort_staging();
        if (rc != TH_OK) return rc;
        TH_HIP(e);
    }
    rc = finish_specs(tm, pending, &fresh, true);
    if (prof) fprintf(stderr, "th_tm_add_tracks prof: staging + launches %.2f ms (of which pyramid / STFT launches %.2f), drain %.2f\n", tp1 - tp0, t_launch, now() - tp1);
    if (rc != TH_OK) {
        abort_staging();
        return rc;
    }
    commit_specs(fresh);  // (into the staged channels)
    for (auto &kv : created) tm->plans[kv.first] = kv.second;
    for (auto &kv : staged) {
        Track &dst = tm->tracks[kv.first];  // re-adding an id replaces it (reload_tracks, core/mod.rs:73-82)
        free_track(dst);
        dst = std::move(kv.second);
        kv.second.d_pool = nullptr;
    }
    std::set<size_t> uniq(added.begin(), added.end());
    tm->no_spec_img_ids.insert(tm->no_spec_img_ids.end(), uniq.begin(), uniq.end());  // core/mod.rs:70
    tm->invalidate_all();                                                             // lib.rs:192
    return writer_done(tm, TH_OK);
    TH_CATCH
}

TH_API int th_tm_add_track(th_tm *tm, size_t id, uint32_t sr, uint32_t n_channels, const float *const *channels,
                           size_t n_samples) {
    return th_tm_add_tracks(tm, 1, &id, &sr, &n_channels, channels, &n_samples);
}

TH_API int th_tm_remove_track(th_tm *tm, size_t id) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    auto it = tm->tracks.find(id);
    if (it == tm->tracks.end()) return fail(TH_ERR_NOT_FOUND, "Track %zu does not exist", id);
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    free_track(it->second);
    tm->tracks.erase(it);
    prune_axis_tabs(tm);
    retain_plans(tm);      // core/mod.rs:96-99
    tm->invalidate_all();  // lib.rs:221
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_apply_track_list_changes(th_tm *tm, size_t *updated_ids, size_t cap, size_t *n_updated,
                                          uint32_t *max_sr) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    std::vector<size_t> upd;
    int rc = writer_done_images(tm, update_spec_imgs(tm, false, &upd));
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
    std::shared_lock<std::shared_mutex> rl(tm->rw);
    if (min_dB) *min_dB = tm->min_dB;
    if (max_dB) *max_dB = tm->max_dB;
    if (max_sr) *max_sr = tm->max_sr;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_spec_shape(const th_tm *tm, size_t id, uint32_t ch, size_t *n_frames, size_t *height) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::shared_lock<std::shared_mutex> rl(tm->rw);
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
    std::shared_lock<std::shared_mutex> rl(tm->rw);
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
    std::unique_lock<std::shared_mutex> wl(tm->rw);  // uses the context stream: exclusive, like a mutator
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
    std::unique_lock<std::shared_mutex> wl(tm->rw);
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

TH_API int th_tm_put_img(th_tm *tm, size_t id, uint32_t ch, const uint16_t *img, size_t height, size_t width) {
    TH_TRY
    TH_REQUIRE(tm && img, "NULL argument");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    Channel *c = find_channel(tm, id, ch);
    if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    if (height != c->img_h || width != c->img_w)
        return fail(TH_ERR_INVALID_ARG, "image is %zu x %zu, the resident one %zu x %zu", height, width, c->img_h, c->img_w);
    if (height && width)
        TH_HIP(hipMemcpy2DAsync(c->d_img, c->img_pitch * sizeof(uint16_t), img, width * sizeof(uint16_t), width * sizeof(uint16_t),
                                height, hipMemcpyHostToDevice, tm->ctx->stream));
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));  // img is the caller's
    std::vector<Channel *> one{c};
    int rc = build_mips(tm, one);
    if (rc != TH_OK) free_mips(*c);
    tm->invalidate_spectrogram();
    return writer_done(tm, rc);
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

TH_API int th_tm_set_lod_source(th_tm *tm, int per_request) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    const int want = per_request ? 1 : 0;
    if (want == tm->lod_source) return TH_OK;
    TH_HIP(hipSetDevice(tm->ctx->device));
    std::vector<Channel *> have;
    for (auto &kv : tm->tracks)
        for (Channel &ch : kv.second.ch)
            if (ch.d_img) have.push_back(&ch);
    // 1: the pyramids go (build_mips frees them); 0: every resident image gets its levels back
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));  // (may fail: the route has not changed yet)
    const int old_source = tm->lod_source;
    tm->lod_source = want;  // build_mips reads it
    int rc = build_mips(tm, have);
    if (rc != TH_OK) {
        // failure-atomic like the other writers: the route stays what it was.  The channels are left without
        // levels — never with stale ones — and are served through the per-request fallback, which is correct on either route.
        for (Channel *ch : have) free_mips(*ch);
        tm->lod_source = old_source;
    }
    prune_axis_tabs(tm);
    return writer_done(tm, rc);
    TH_CATCH
}

// get_spectrogram_tile — lib.rs:369-389 → RenderTileCache::spectrogram_tile (render_tiles.rs:171-188).
// Level (0, 0): a crop of the u16 image; LOD > 0: a crop of the pre-built mip level — either way one
// raster launch on the request's own stream, one copy into pinned memory, no table upload, no lock shared with other
// requests.
TH_API int th_tm_get_spectrogram_tile(th_tm *tm, size_t id, uint32_t ch, uint32_t level_x, uint32_t level_y,
                                      uint32_t tile_x, uint32_t tile_y, uint8_t *out, size_t cap, size_t *out_len) {
    TH_TRY
    TH_REQUIRE(tm && out && out_len, "NULL argument");
    std::shared_lock<std::shared_mutex> rl(tm->rw);
    Channel *c = find_channel(tm, id, ch);
    if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    const uint64_t revision = tm->spectrogram_revision();
    const uint16_t *src = c->d_img;
    uint32_t src_w = (uint32_t)c->img_w, src_h = (uint32_t)c->img_h, src_pitch = (uint32_t)c->img_pitch;
    if (level_x != 0 || level_y != 0) {
        auto im = c->mips.find({level_x, level_y});
        if (tm->lod_source != 0 || im == c->mips.end()) {
            // per-request resampling of the crop box from the level-0 image (the reference's own flow): the
            // context-stream path, serialised by the context mutex
            return th_encode_spectrogram_tile_dev(tm->ctx, c->d_img, c->img_h, c->img_w, c->img_pitch,
                                                  tm->colormap_rgba.data(), tm->colormap_rgba.size(), revision, level_x,
                                                  level_y, tile_x, tile_y, out, cap, out_len);
        }
        src = im->second.d;
        src_w = im->second.w;
        src_h = im->second.h;
        src_pitch = im->second.pitch;
    }
    const TileGeom g = spectrogram_tile_geometry(c->img_w, c->img_h, level_x, level_y, tile_x, tile_y);
    const size_t px_bytes = g.width * g.height * 4, need = 40 + px_bytes;
    *out_len = need;
    if (cap < need) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", need);
    put_u64(out, revision);
    put_u32(out + 8, (uint32_t)g.width);
    put_u32(out + 12, (uint32_t)g.height);
    put_u32(out + 16, level_x);
    put_u32(out + 20, level_y);
    put_u32(out + 24, tile_x);
    put_u32(out + 28, tile_y);
    put_u32(out + 32, (uint32_t)g.origin_x);
    put_u32(out + 36, (uint32_t)g.origin_y);
    if (g.width == 0 || g.height == 0) return TH_OK;
    if (g.lod_w != src_w || g.lod_h != src_h) return fail(TH_ERR_INTERNAL, "mip level (%u, %u) has the wrong shape", level_x, level_y);
    if (px_bytes > TILE_BYTES_MAX) return fail(TH_ERR_INTERNAL, "tile larger than a reader slot");
    if (!tm->d_colormap) {  // no th_tm_set_colormap yet: the default two-colour map
        rl.unlock();
        {
            std::unique_lock<std::shared_mutex> wl(tm->rw);
            TH_HIP(hipSetDevice(tm->ctx->device));
            if (!tm->d_colormap) {
                int rc = upload_colormap(tm);
                if (rc != TH_OK) return rc;
            }
        }
        return th_tm_get_spectrogram_tile(tm, id, ch, level_x, level_y, tile_x, tile_y, out, cap, out_len);
    }
    TH_HIP(hipSetDevice(tm->ctx->device));
    SlotLease lease{tm, nullptr};
    int rc = acquire_slot(tm, &lease.slot);
    if (rc != TH_OK) return rc;
    ReaderSlot &sl = *lease.slot;
    // (a mip level may still be building behind the mutator that re-made the images: th_tm::mips_ev)
    // (asked of the host first: a wait packet for an event that completed long ago still costs a request ~20 us)
    if ((level_x != 0 || level_y != 0) && tm->mips_ev != nullptr && hipEventQuery(tm->mips_ev) != hipSuccess) {
        (void)hipGetLastError();  // (hipErrorNotReady is the answer, not a failure)
        TH_HIP(hipStreamWaitEvent(sl.stream, tm->mips_ev, 0));
    }
    // The raster kernel writes the pixels straight into the slot's pinned, device-visible host buffer (16 bytes per lane,
    // 1 KB per wave-instruction over PCIe): no device staging tile, no copy-engine transfer queued behind other readers'.
    TH_HIP(launch_raster_tile(src, src_w, src_h, src_pitch, (uint32_t)g.origin_x, (uint32_t)g.origin_y, (uint32_t)g.width,
                              (uint32_t)g.height, sl.h_tile_dev, tm->d_colormap, (uint32_t)(tm->colormap_rgba.size() / 4),
                              sl.stream));
    TH_HIP(hipStreamSynchronize(sl.stream));
    std::memcpy(out + 40, sl.h_tile, px_bytes);
    return TH_OK;
    TH_CATCH
}

// get_waveform_tile — lib.rs:342-367: cache lookup (:350-355); on a miss the tile's bins are a contiguous piece of the
// channel's resident pyramid level (built once at add_tracks from one pass over the audio): one copy, no kernel.
TH_API int th_host_alloc(th_ctx *c, size_t bytes, void **ptr) {
    TH_TRY
    TH_REQUIRE(c && ptr, "NULL argument");
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(hipHostMalloc(ptr, std::max<size_t>(bytes, 1), hipHostMallocMapped | hipHostMallocPortable));
    return TH_OK;
    TH_CATCH
}
TH_API int th_host_free(th_ctx *c, void *ptr) {
    TH_TRY
    TH_REQUIRE(c, "ctx is NULL");
    if (!ptr) return TH_OK;
    TH_HIP(hipSetDevice(c->device));
    TH_HIP(hipHostFree(ptr));
    return TH_OK;
    TH_CATCH
}

// N tiles, one raster launch (th_raster_tiles_dev on the context stream), one wait.  Level (0, 0) tiles are crops of the
// u16 image, LOD > 0 tiles crops of their mip level; requests for levels the pyramid does not hold (or with the per-request
// route selected) go through the single-tile path one by one.
TH_API int th_tm_get_spectrogram_tiles(th_tm *tm, const th_tile_request *reqs, size_t n, uint8_t *out, size_t cap,
                                       size_t *offsets, size_t *out_len) {
    TH_TRY
    TH_REQUIRE(tm && out_len && (n == 0 || (reqs && offsets)), "NULL argument");
    *out_len = 0;
    if (n == 0) {
        if (offsets) offsets[0] = 0;
        return TH_OK;
    }
    std::shared_lock<std::shared_mutex> rl(tm->rw);
    if (!tm->d_colormap) {  // no th_tm_set_colormap yet: upload the default map (as the single-tile path does)
        rl.unlock();
        {
            std::unique_lock<std::shared_mutex> wl(tm->rw);
            TH_HIP(hipSetDevice(tm->ctx->device));
            if (!tm->d_colormap) {
                int rc = upload_colormap(tm);
                if (rc != TH_OK) return rc;
            }
        }
        rl.lock();
    }
    const uint64_t revision = tm->spectrogram_revision();
    struct Item {
        TileGeom g;
        const uint16_t *src;
        uint32_t src_w, src_h, src_pitch;
        bool single;  // served by th_tm_get_spectrogram_tile
    };
    std::vector<Item> items(n);
    size_t total = 0;
    for (size_t i = 0; i < n; i++) {
        const th_tile_request &r = reqs[i];
        Channel *c = find_channel(tm, r.id, r.ch);
        if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", r.id, r.ch);
        Item &it = items[i];
        it.g = spectrogram_tile_geometry(c->img_w, c->img_h, r.level_x, r.level_y, r.tile_x, r.tile_y);
        it.src = c->d_img;
        it.src_w = (uint32_t)c->img_w;
        it.src_h = (uint32_t)c->img_h;
        it.src_pitch = (uint32_t)c->img_pitch;
        it.single = false;
        if (r.level_x != 0 || r.level_y != 0) {
            auto im = c->mips.find({r.level_x, r.level_y});
            if (tm->lod_source != 0 || im == c->mips.end()) {
                it.single = true;
            } else {
                it.src = im->second.d;
                it.src_w = im->second.w;
                it.src_h = im->second.h;
                it.src_pitch = im->second.pitch;
                if (it.g.width && it.g.height && (it.g.lod_w != it.src_w || it.g.lod_h != it.src_h))
                    return fail(TH_ERR_INTERNAL, "mip level (%u, %u) has the wrong shape", r.level_x, r.level_y);
            }
        }
        offsets[i] = total;
        total += (40 + it.g.width * it.g.height * 4 + 63) / 64 * 64;
    }
    offsets[n] = total;
    *out_len = total;
    if (cap < total || !out) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", total);
    TH_HIP(hipSetDevice(tm->ctx->device));
    // where the kernel writes: the caller's buffer when the GPU can reach it, else the manager's pinned staging buffer
    uint8_t *dev_base = nullptr;
    {
        hipPointerAttribute_t at{};
        void *dp = nullptr;
        const hipError_t qe = hipPointerGetAttributes(&at, out);
        if (qe == hipSuccess && at.type == hipMemoryTypeDevice)
            return fail(TH_ERR_INVALID_ARG, "out is device memory: th_tm_get_spectrogram_tiles writes tile headers from the host");
        if (qe == hipSuccess && at.type == hipMemoryTypeHost && hipHostGetDevicePointer(&dp, out, 0) == hipSuccess && dp)
            dev_base = static_cast<uint8_t *>(dp);
        else
            (void)hipGetLastError();  // (pageable memory: the query fails, that is the answer)
    }
    std::unique_lock<std::mutex> bl(tm->batch_mu, std::defer_lock);
    const bool staged = dev_base == nullptr;
    if (staged) {
        bl.lock();
        if (tm->batch_stage_cap < total) {
            if (tm->batch_stage) (void)hipHostFree(tm->batch_stage);
            tm->batch_stage = tm->batch_stage_dev = nullptr;
            tm->batch_stage_cap = 0;
            TH_HIP(hipHostMalloc(&tm->batch_stage, total, hipHostMallocMapped));
            TH_HIP(hipHostGetDevicePointer(&tm->batch_stage_dev, tm->batch_stage, 0));
            tm->batch_stage_cap = total;
        }
        dev_base = static_cast<uint8_t *>(tm->batch_stage_dev);
    }
    std::vector<th_raster_desc> descs;
    descs.reserve(n);
    for (size_t i = 0; i < n; i++) {
        const Item &it = items[i];
        if (it.single || !it.g.width || !it.g.height) continue;
        th_raster_desc d{};
        d.img = it.src;
        d.rgba = dev_base + offsets[i] + 40;
        d.img_width = it.src_w;
        d.img_height = it.src_h;
        d.origin_x = (uint32_t)it.g.origin_x;
        d.origin_y = (uint32_t)it.g.origin_y;
        d.width = (uint32_t)it.g.width;
        d.height = (uint32_t)it.g.height;
        d.img_pitch = it.src_pitch;
        descs.push_back(d);
    }
    if (!descs.empty()) {
        std::lock_guard<std::recursive_mutex> lk(tm->ctx->mu);  // the context stream: one user at a time
        int rc = th_raster_tiles_dev(tm->ctx, descs.data(), descs.size(), tm->d_colormap, (uint32_t)(tm->colormap_rgba.size() / 4));
        if (rc != TH_OK) return rc;
        TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    }
    if (staged) {
        // the staging buffer is held only until its pixels are copied out: the per-request fallbacks and the
        // headers below no longer keep other pageable-buffer callers waiting
        for (size_t i = 0; i < n; i++) {
            const Item &it = items[i];
            const size_t px = it.g.width * it.g.height * 4;
            if (!it.single && px) std::memcpy(out + offsets[i] + 40, static_cast<uint8_t *>(tm->batch_stage) + offsets[i] + 40, px);
        }
        bl.unlock();
    }
    for (size_t i = 0; i < n; i++) {
        const th_tile_request &r = reqs[i];
        const Item &it = items[i];
        uint8_t *rec = out + offsets[i];
        if (it.single) {
            size_t len = 0;
            const Channel *c = find_channel(tm, r.id, r.ch);  // (checked in the first pass; the shared lock is still held)
            int rc = th_encode_spectrogram_tile_dev(tm->ctx, c->d_img, c->img_h, c->img_w, c->img_pitch, tm->colormap_rgba.data(),
                                                    tm->colormap_rgba.size(), revision, r.level_x, r.level_y, r.tile_x, r.tile_y,
                                                    rec, offsets[i + 1] - offsets[i], &len);
            if (rc != TH_OK) return rc;
            continue;
        }
        put_u64(rec, revision);
        put_u32(rec + 8, (uint32_t)it.g.width);
        put_u32(rec + 12, (uint32_t)it.g.height);
        put_u32(rec + 16, r.level_x);
        put_u32(rec + 20, r.level_y);
        put_u32(rec + 24, r.tile_x);
        put_u32(rec + 28, r.tile_y);
        put_u32(rec + 32, (uint32_t)it.g.origin_x);
        put_u32(rec + 36, (uint32_t)it.g.origin_y);
    }
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_get_waveform_tile(th_tm *tm, size_t id, uint32_t ch, uint32_t level, uint32_t tile_index,
                                   uint8_t *out, size_t cap, size_t *out_len) {
    TH_TRY
    TH_REQUIRE(tm && out && out_len, "NULL argument");
    std::shared_lock<std::shared_mutex> rl(tm->rw);
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
    size_t start, bins, spb;
    waveform_tile_geometry(c->n, level, tile_index, &start, &bins, &spb);
    const size_t need = 24 + bins * 12;
    *out_len = need;
    if (cap < need) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu bytes", need);
    put_u64(out, revision);
    put_u32(out + 8, (uint32_t)bins);
    put_u32(out + 12, spb > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)spb);
    put_u32(out + 16, tile_index);
    put_u32(out + 20, 0);
    if (bins) {
        if (!c->d_wav || !c->pyr_levels) return fail(TH_ERR_INTERNAL, "channel %zu_%u has no waveform pyramid", id, ch);
        // a level above the pyramid's last one still has exactly one bin, over the same samples
        const uint32_t lv = level < c->pyr_levels ? level : c->pyr_levels - 1;
        const size_t first_bin = lv == level ? (size_t)tile_index * 1024 : 0;
        TH_HIP(hipSetDevice(tm->ctx->device));
        SlotLease lease{tm, nullptr};
        int rc = acquire_slot(tm, &lease.slot);
        if (rc != TH_OK) return rc;
        ReaderSlot &sl = *lease.slot;
        if (lv < PYR_FIRST) {
            // levels 0 and 1 are not materialised: copy the tile's samples (4 / 8 KB against 12 KB of bins) and form the bins
            // here exactly as waveform_bin_stats does for short slices (render_tiles.rs:261-279: f32::min / f32::max folds,
            // sequential sum / len — one or two samples per bin)
            const size_t spb_lv = (size_t)1 << lv;
            const size_t s0 = lv == level ? start : 0, s1 = std::min<size_t>(c->n, s0 + bins * spb_lv);
            TH_HIP(hipMemcpyAsync(sl.h_tile, c->d_wav + s0, (s1 - s0) * 4, hipMemcpyDeviceToHost, sl.stream));
            TH_HIP(hipStreamSynchronize(sl.stream));
            const float *x = reinterpret_cast<const float *>(sl.h_tile);
            for (size_t b = 0; b < bins; b++) {
                const size_t a = b * spb_lv, e = std::min(s1 - s0, a + spb_lv);
                float mn = INFINITY, mx = -INFINITY, sum = 0.0f;
                for (size_t i = a; i < e; i++) {
                    mn = std::fmin(mn, x[i]);
                    mx = std::fmax(mx, x[i]);
                    sum = sum + x[i];
                }
                const float mean = sum / (float)(e - a);
                std::memcpy(out + 24 + 12 * b, &mn, 4);
                std::memcpy(out + 28 + 12 * b, &mx, 4);
                std::memcpy(out + 32 + 12 * b, &mean, 4);
            }
        } else {
            const float *d_src = c->d_pyr + (th_waveform_pyramid_offset(c->n, lv) - th_waveform_pyramid_offset(c->n, PYR_FIRST)) + 3 * first_bin;
            TH_HIP(hipMemcpyAsync(sl.h_tile, d_src, bins * 12, hipMemcpyDeviceToHost, sl.stream));
            TH_HIP(hipStreamSynchronize(sl.stream));
            std::memcpy(out + 24, sl.h_tile, bins * 12);
        }
    }
    tm->cache.store(id, ch, revision, level, tile_index, out, need);
    return TH_OK;
    TH_CATCH
}

// get_audio_render_metadata — lib.rs:321-340 → RenderTileCache::metadata (render_tiles.rs:101-122)
TH_API int th_tm_get_audio_render_metadata(th_tm *tm, size_t id, uint32_t ch, double track_sec, int is_clipped,
                                           th_render_metadata *out) {
    TH_TRY
    TH_REQUIRE(tm && out, "NULL argument");
    std::shared_lock<std::shared_mutex> rl(tm->rw);
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

// device pointer + shape of one level of a channel's LOD mip pyramid ((0, 0) = the image itself): parity tests
TH_API int th_tm_lod_footprint(th_tm *tm, size_t *n_axis_tables, size_t *axis_table_bytes, size_t *mip_bytes) {
    TH_TRY
    TH_REQUIRE(tm, "tm is NULL");
    std::shared_lock<std::shared_mutex> rl(tm->rw);
    size_t ab = 0, mb = 0;
    for (auto &kv : tm->axis_tabs) ab += kv.second.bytes;
    for (auto &kv : tm->tracks)
        for (Channel &ch : kv.second.ch) mb += ch.mips_elems * sizeof(uint16_t);
    if (n_axis_tables) *n_axis_tables = tm->axis_tabs.size();
    if (axis_table_bytes) *axis_table_bytes = ab;
    if (mip_bytes) *mip_bytes = mb;
    return TH_OK;
    TH_CATCH
}

TH_API int th_tm_mip_level(th_tm *tm, size_t id, uint32_t ch, uint32_t level_x, uint32_t level_y, uint16_t *out, size_t cap,
                           size_t *width, size_t *height) {
    TH_TRY
    TH_REQUIRE(tm, "NULL argument");
    std::unique_lock<std::shared_mutex> wl(tm->rw);
    TH_HIP(hipSetDevice(tm->ctx->device));
    Channel *c = find_channel(tm, id, ch);
    if (!c || !c->d_img) return fail(TH_ERR_NOT_FOUND, "Spectrogram %zu_%u does not exist", id, ch);
    const uint16_t *src = c->d_img;
    size_t w = c->img_w, h = c->img_h, pitch = c->img_pitch;
    if (level_x || level_y) {
        auto im = c->mips.find({level_x, level_y});
        if (im == c->mips.end()) return fail(TH_ERR_NOT_FOUND, "mip level (%u, %u) is not resident", level_x, level_y);
        src = im->second.d;
        w = im->second.w;
        h = im->second.h;
        pitch = im->second.pitch;
    }
    if (width) *width = w;
    if (height) *height = h;
    if (!out) return TH_OK;
    if (cap < w * h) return fail(TH_ERR_BUFFER_TOO_SMALL, "need %zu pixels", w * h);
    if (w && h)
        TH_HIP(hipMemcpy2DAsync(out, w * sizeof(uint16_t), src, pitch * sizeof(uint16_t), w * sizeof(uint16_t), h,
                                hipMemcpyDeviceToHost, tm->ctx->stream));
    TH_HIP(hipStreamSynchronize(tm->ctx->stream));
    return TH_OK;
    TH_CATCH
}
