// context.h — definitions of the opaque handles (th_ctx, th_plan) shared by api.hip and
// track_manager.hip.
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <vector>

#include "common.h"
#include "stft_core.h"

namespace th {

// Small device buffer holding a descriptor table; remembers the last uploaded bytes so that
// re-sending an identical table (bench loops, repeated tile requests) costs nothing.
struct DeviceTable {
    void *dptr = nullptr;
    size_t cap = 0;
    std::vector<unsigned char> last;
    int ensure(size_t bytes);
    int upload(hipStream_t s, const void *src, size_t bytes);
    void release();
};

}  // namespace th

struct th_ctx {
    int device = 0;
    uint32_t n_cu = 256;  // compute units (persistent-grid size of the wave kernel)
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    // Serialises use of the stream-side scratch below; recursive so composed entry points
    // (tile encoders → batched launchers) can hold it across the whole request.
    std::recursive_mutex mu;
    th::DeviceTable img_jobs, img_start, raster_jobs, raster_start, wave_jobs, wave_start, colormap, tile_out, lod_tabs, lod_tmp,
        pyr_jobs, pyr_sums, fused_jobs, fused_start, fused_ptrs;
    // the descriptor batches the img / raster tables were built from (identical batch -> tables reused as they are)
    std::vector<unsigned char> img_descs_key, raster_descs_key, fused_key;
    uint32_t img_tiles_key = 0, raster_blocks_key = 0, fused_blocks_key = 0;
    void release_scratch() {
        img_descs_key.clear();
        raster_descs_key.clear();
        fused_key.clear();
        fused_jobs.release();
        fused_start.release();
        fused_ptrs.release();
        img_jobs.release();
        img_start.release();
        raster_jobs.release();
        raster_start.release();
        wave_jobs.release();
        wave_start.release();
        colormap.release();
        tile_out.release();
        lod_tabs.release();
        lod_tmp.release();
        pyr_jobs.release();
        pyr_sums.release();
    }
};

struct th_plan {
    th_ctx *ctx = nullptr;
    uint32_t sr = 0;
    int freq_scale = 0;
    int kernel_choice = 0;  // 0 auto, 1 generic, 2 wave, 3 wave with the matrix-core mel kernel (no fused epilogue), 4 wave without the phased mode
    int wave_waves = 0;     // tuning: waves per workgroup of the wave kernel (0 = default)
    int wave_chunk = 0;     // tuning: frames per chunk of the wave kernel (0 = default)
    // th_plan_time_kernel: a ring of event pairs around the STFT kernel launch (no synchronisation while recording)
    static constexpr size_t TIMER_SLOTS = 64;
    bool time_kernel = false;
    uint64_t timed_launches = 0;
    std::vector<hipEvent_t> ev_k0, ev_k1;
    th::StftGeom g{};
    float *d_window = nullptr;
    th::cf32 *d_tw = nullptr;
    // Bluestein plans (an odd factor of n_fft above 63; stft_bluestein_kernel): complex-double tables, NULL otherwise
    double *d_bs_chirp = nullptr, *d_bs_bhat = nullptr, *d_bs_twm = nullptr, *d_bs_tws = nullptr;
    bool bluestein() const { return d_bs_chirp != nullptr; }
    uint32_t *d_queue_head = nullptr;  // wave kernel: chunk queue head (rewound by wave_post_kernel after every launch)
    bool queue_dirty = false;          // a wave launch went out whose rewind did not: the next launch zeroes the head first
    th::cf32 *d_wtab = nullptr;  // wave kernel: 0.5 * zero-padded window as (even, odd) pairs
    th::cf32 *d_wtab_phased = nullptr;  // phased mode: 48 zero pairs + the table with the window at offset 0 (NULL: not applicable)
    bool use_wave() const;
    float *d_mel_fb = nullptr;
    uint32_t *d_mel_lo = nullptr, *d_mel_hi = nullptr;
    std::vector<float> h_mel_fb;
    // MFMA mel path: filterbank packed per (N tile of 16 mels, K block of 16 bins) in operand order (256 floats per
    // block, band blocks only, + one all-zero block), per-tile band {klo, khi, first block}
    float *d_mel_bt = nullptr;
    uint32_t *d_mel_band = nullptr, *d_mel_slice = nullptr;  // + slices of the tile range with ~equal K-group counts
    uint32_t mel_kblocks = 0, mel_ntiles = 0, mel_zero_block = 0, mel_slices = 0;
    uint32_t *d_mel_rows = nullptr;  // short rows under narrow filters: the per-mel table of mel_rows_kernel (kernels.h)
    uint32_t mel_rows_groups = 0;
    th::DeviceTable amp_buf, mel_jobs, mel_tile_start;  // amplitude scratch + job tables of mel_mfma_kernel
    size_t amp_zeroed = 0;                               // bytes of amp_buf known to be zero-initialised
    th::DeviceTable chunk_mm;                            // (min, max) per chunk of the wave kernel's last launch
    th::DeviceTable gen_scratch;                         // n_fft >= 32768: frame buffers of the generic kernel (global scratch)
    th::DeviceTable post_jobs;                           // per-channel tile ranges for wave_post_kernel
    bool use_mel_mfma() const;   // mel plan: amplitude rows + mel_mfma_kernel
    bool use_mel_fused() const;  // mel plan: filterbank fused into the wave kernel's epilogue (mel_fuse.h)
    bool use_mel_moment_small() const;  // n_fft 1024 / 2048 mel plan whose table forms do not fit LDS: the moment-form epilogue (round 6)
    int long_plan() const;       // n_fft 8192 .. 65536: 0 = the size's default plan, 1 = stft_block_kernel, 2 = stft_subwave_kernel (WaveOut::long_plan)
    // fused mel epilogue: device copy of the mel_fuse.h word table
    uint32_t *d_mel_fuse = nullptr;
    uint32_t mel_fuse_words = 0, mel_fuse_slots = 0, mel_fuse_groups = 0;
    // the same epilogue as banded sums, lane = mel (build_mel_band): the default where its table fits; selector 8 keeps the pieces
    uint32_t *d_mel_bsum = nullptr;
    uint32_t mel_bsum_words = 0, mel_bsum_groups = 0, mel_bsum_hdr[16] = {};  // (header: offset and taps per group)
    th::cf32 *d_twc = nullptr;  // n_fft 32768: the combining pass's per-thread constants (kernels_stft_long.hip)
    uint32_t mel_bsum_reach = 0;  // one past the highest amplitude index the banded sums read (MelBandHost::reach)
    // n_fft 4096 mel plans (round 6): the moment form of the filterbank in the FFT kernel's epilogue (build_mel_moments, mel_fuse.h)
    uint32_t *d_mel_mom = nullptr;
    uint32_t mel_mom_groups = 0, mel_mom_taps = 0;
    double mel_mom_max_dev = 0.0, mel_mom_max_amp = 0.0;
    bool mel_bsum_fits() const;
    th::DeviceTable jobs, tile_start;            // main launch: jobs + first chunk of every job (generic kernel) or the
                                                 // per-chunk (job, first frame) table (wave kernels)
    th::DeviceTable edge_jobs, edge_tile_start;  // boundary frames handed to the generic kernel
};
