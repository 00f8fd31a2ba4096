// Native frame driver (gfx950): ImportanceRenderer.forward (renderer.py:286-398) enqueued from C++ on two HIP streams.
//
// Why native: a frame is ~75 kernel launches of 5-800 us each; issued one by one through the Python binding the host
// needs about as long to enqueue a frame as the GPU needs to run it, and the ray side of the frame starts only after
// the whole encoder chain has been queued.  Here the launch order interleaves the two streams so that both start at
// once: SMPL tables (side) -> cell lists / sampling / table folds (main) -> voxel encoder (side) -> warp, gather, MLP,
// compositing (main).
#include "common.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <functional>
#include <mutex>

int sherf_svox_encode_impl(const sherf_svox_plan* p, const int32_t* coord, const float* feat, int n, int training,
                           sherf_vox_level* levels_out_host, sherf_stream_t stream, hipEvent_t ev, int ev_layer,
                           sherf_stream_t aux, hipEvent_t* lev_ev, const std::function<int()>* after_levels, int fold_half);   // svox.hip
int sherf_sample_mask_nn_impl(const float* ray_o, const float* ray_d, const float* near, const float* far, int R, int S, const float* Rg, const float* Th,
                              const float* grid_hdr, const int32_t* cell_start, const float* cell_pts, const uint32_t* near_mask, int64_t capacity,
                              int32_t* counters, int32_t* ray_base, int32_t* ray_cnt, int32_t* cs_idx, int32_t* cs_vid, float* cs_xs, int32_t* dense_vid,
                              uint64_t* ray_mask, int32_t* scan_ws, const int32_t* near_hdr, const uint16_t* near_list, sherf_stream_t stream,
                              int32_t* sticky);                                                                                 // sample.hip

namespace {

constexpr int kMaxDev = 16;
constexpr int kRing = 64;
constexpr int kMaxParts = 8;

struct DevState {
    bool init = false;
    hipEvent_t ev_start, ev_smpl, ev_enc, ev_mid, ev_fold, ev_lev[8], ev_cnt, ev_part[kMaxParts], ev_mlp;
    int32_t* host_nv = nullptr;      // pinned: the frame's valid-sample count for SHERF_FRAME_EXACT_GRIDS
    hipEvent_t ev_rep[8];            // SHERF_FRAME_REPORT_COUNT: a ring of (pinned word, event) pairs; host_nv[8 + slot]
    int rep_next = 0;
    unsigned long long rep_seq[8] = {0, 0, 0, 0, 0, 0, 0, 0}, rep_counter = 0;   // which frame holds a slot: a reader whose frame has been overtaken by eight others is told so
    hipStream_t cap_stream = nullptr;   // frame graphs are CAPTURED on this library-owned non-blocking stream (the caller's may be the legacy default
                                        // stream, which cannot capture) and LAUNCHED on the caller's
};
// the REPORT_COUNT slot of the last frame THIS THREAD enqueued (per device): another thread's (renderer's) frames on the same device take
// their own slots, and a reader never waits with the enqueue lock held
thread_local int t_rep_last[kMaxDev] = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1};
thread_local unsigned long long t_rep_seq[kMaxDev] = {0};
DevState g_dev[kMaxDev];
std::mutex g_mu;          // profiling ring + event creation
std::mutex g_frame_mu;    // one enqueue at a time: the join events are shared per device

// frame profiling ring: kEv timing events per frame + the host time of the enqueue
constexpr int kEv = 7;     // 0 start (main) 1 SMPL tables done (side) 2 encoder done (side) 3 main reaches the encoder join
                           // 4 gather done 5 MLP done 6 compositing done
bool g_prof_on = false;
int g_prof_n = 0;
hipEvent_t g_prof_ev[kRing][kEv];
float g_prof_host_ms[kRing];
bool g_prof_init = false;

}  // namespace

extern "C" int sherf_struct_sizes(int32_t* sizes_host, int32_t n) {
    SHERF_CHECK_ARG(sizes_host && n >= 5);
    sizes_host[0] = (int32_t)sizeof(sherf_vox_level);
    sizes_host[1] = (int32_t)sizeof(sherf_svox_level_ws);
    sizes_host[2] = (int32_t)sizeof(sherf_svox_layer);
    sizes_host[3] = (int32_t)sizeof(sherf_svox_plan);
    sizes_host[4] = (int32_t)sizeof(sherf_frame);
    return SHERF_OK;
}

extern "C" int sherf_profile_frames(int enable) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (enable && !g_prof_init) {
        for (int i = 0; i < kRing; ++i)
            for (int j = 0; j < kEv; ++j) SHERF_HIP_CHECK(hipEventCreate(&g_prof_ev[i][j]));
        g_prof_init = true;
    }
    g_prof_on = enable != 0;
    g_prof_n = 0;
    return SHERF_OK;
}

extern "C" int sherf_profile_frames_read(float* ms_host, int32_t max_n, int32_t* n_host) {
    SHERF_CHECK_ARG(ms_host && n_host && max_n > 0);
    std::lock_guard<std::mutex> lk(g_mu);
    const int have = g_prof_n < kRing ? g_prof_n : kRing;
    const int n = have < max_n ? have : max_n;
    const int first = g_prof_n - have;
    for (int i = 0; i < n; ++i) {
        const int slot = (first + i) % kRing;
        float* o = ms_host + (size_t)i * SHERF_PROF_FIELDS;
        SHERF_HIP_CHECK(hipEventSynchronize(g_prof_ev[slot][kEv - 1]));
        o[0] = g_prof_host_ms[slot];
        for (int j = 1; j < kEv; ++j) SHERF_HIP_CHECK(hipEventElapsedTime(&o[j], g_prof_ev[slot][0], g_prof_ev[slot][j]));
        o[7] = o[5] - o[4];
    }
    *n_host = n;
    return SHERF_OK;
}

extern "C" int sherf_frame_count(int32_t* nv_host) {
    SHERF_CHECK_ARG(nv_host);
    int dev = 0;
    SHERF_HIP_CHECK(hipGetDevice(&dev));
    SHERF_CHECK_ARG(dev >= 0 && dev < kMaxDev);
    DevState& d = g_dev[dev];
    const int slot = t_rep_last[dev];
    hipEvent_t ev;
    auto overtaken = [&]() -> int {
        snprintf(g_sherf_err, sizeof(g_sherf_err), "sherf_frame_count: eight later frames on this device have reused the slot of this thread's last frame");
        return SHERF_EINVAL;
    };
    {
        std::lock_guard<std::mutex> frame_lock(g_frame_mu);
        SHERF_CHECK_ARG(d.init && slot >= 0);
        if (d.rep_seq[slot] != t_rep_seq[dev]) return overtaken();
        ev = d.ev_rep[slot];
    }
    SHERF_HIP_CHECK(hipEventSynchronize(ev));          // (no lock held: other threads keep enqueueing; the ring has 8 slots per device)
    const int32_t nv = d.host_nv[8 + slot];
    {
        std::lock_guard<std::mutex> frame_lock(g_frame_mu);     // the word is this frame's only if nobody took the slot while we waited
        if (d.rep_seq[slot] != t_rep_seq[dev]) return overtaken();
    }
    *nv_host = nv;
    return SHERF_OK;
}

// The frame's launches, enqueued one by one on the three streams (with g_frame_mu held).  sherf_render_frame below either calls this, or captures
// what it enqueues into a hipGraph once and replays the graph from then on.
static int render_frame_enqueue(const sherf_frame* f, int phase, sherf_vox_level* levels, sherf_stream_t stream_main,
                                sherf_stream_t stream_side, sherf_stream_t stream_aux) {
    hipStream_t main = as_stream(stream_main), side = as_stream(stream_side);
    if (phase == 4) {                                                // the sampler alone: counters[0] = the frame's valid samples
        SHERF_CHECK_ARG(f->R > 0 && f->S > 0 && f->capacity > 0);
        const bool lists = f->near_hdr && f->near_list;
        SHERF_RUN(sherf_build_cells2(f->verts, f->Rg, f->Th, f->tverts, SHERF_V, 0.05f, f->grid_hdr, f->cell_start, f->cell_pts,
                                     f->cell_scratch, lists ? nullptr : f->near_mask, stream_main));
        if (lists)
            SHERF_RUN(sherf_build_near_lists(f->grid_hdr, f->cell_pts, SHERF_V, 0.05f, f->near_hdr, f->near_list, f->near_list_cap, f->near_mask, stream_main));
        SHERF_RUN(sherf_sample_mask_nn_impl(f->ray_o, f->ray_d, f->near, f->far, f->R, f->S, f->Rg, f->Th, f->grid_hdr, f->cell_start,
                                       f->cell_pts, f->near_mask, f->capacity, f->counters, f->ray_base, f->ray_cnt, f->cs_idx,
                                       f->cs_vid, f->cs_xs, f->dense_vid, f->ray_mask, f->scan_ws, f->near_hdr, f->near_list, stream_main, f->sticky));
        return SHERF_OK;
    }
    // token-side capacity: what geom / tokens / extras / sample_out hold
    const int64_t tok_cap = (f->tok_capacity > 0 && f->tok_capacity < f->capacity) ? f->tok_capacity : f->capacity;
    const auto host_t0 = std::chrono::steady_clock::now();
    static int cur_slot = -1;          // guarded by g_frame_mu; phase 2 of a split call reuses phase 1's slot
#define SHERF_PROF(j, strm) do { if (cur_slot >= 0) SHERF_HIP_CHECK(hipEventRecord(g_prof_ev[cur_slot][j], strm)); } while (0)
    if (phase & 1) {
        int dev = 0;
        SHERF_HIP_CHECK(hipGetDevice(&dev));
        SHERF_CHECK_ARG(dev >= 0 && dev < kMaxDev);
        DevState& d = g_dev[dev];
        {
            std::lock_guard<std::mutex> lk(g_mu);
            if (!d.init) {
                // (device-to-device events; the two the HOST waits on -- ev_cnt, ev_rep -- keep the default system-scope release)
                const unsigned evf = hipEventDisableTiming | ((sherf_experiment() & 4) ? hipEventReleaseToDevice : 0u);
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_start, evf));
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_smpl, evf));
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_enc, evf));
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_mid, evf));
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_fold, evf));
                for (int k = 0; k < 8; ++k) SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_lev[k], evf));
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_cnt, hipEventDisableTiming));
                for (int k = 0; k < kMaxParts; ++k) SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_part[k], evf));
                SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_mlp, evf));
                SHERF_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&d.host_nv), 64, 0));
                for (int k = 0; k < 8; ++k) SHERF_HIP_CHECK(hipEventCreateWithFlags(&d.ev_rep[k], hipEventDisableTiming));
                d.init = true;
            }
        }
        SHERF_CHECK_ARG(f->R > 0 && f->S > 0 && f->capacity > 0 && f->vox_plan);
        const int V = SHERF_V;
        {
            std::lock_guard<std::mutex> lk(g_mu);
            cur_slot = g_prof_on ? g_prof_n % kRing : -1;
        }
        const int xp = sherf_experiment();
        SHERF_HOST_STAMP(xp, "frame enter");
        // inputs were produced on the caller's stream
        SHERF_PROF(0, main);
        SHERF_GPU_STAMP(xp, main, "frame start");
        SHERF_CAP_TRACE("record ev_start");
        SHERF_HIP_CHECK(hipEventRecord(d.ev_start, main));
        SHERF_CAP_TRACE("side waits ev_start");
        SHERF_HIP_CHECK(hipStreamWaitEvent(side, d.ev_start, 0));
        // ---- a7-a9 per-frame SMPL tables: first needed by the warp (after sampling), so with an aux stream they queue there
        // behind the table folds and level builds and the encoder chain starts at once on the side stream ----
        auto smpl_tables = [&](sherf_stream_t st) -> int {
            SHERF_RUN(sherf_smpl_bones(f->poses, f->shapes, 3, f->J_template, f->J_shapedirs, f->parents, f->A, f->posefeat, st));
            SHERF_RUN(sherf_smpl_offsets(f->posedirs, f->shapedirs, f->posefeat, f->shapes, 3, f->PO, f->SO, st));
            const float *A0 = f->A, *A1 = f->A + 24 * 12, *A2 = f->A + 2 * 24 * 12;
            const float *PO0 = f->PO, *PO1 = f->PO + V * 3, *PO2 = f->PO + 2 * V * 3;
            const float *SO0 = f->SO, *SO2 = f->SO + 2 * V * 3;
            SHERF_RUN(sherf_smpl_t2c_table(f->weights, A0, A1, PO0, SO0, PO1, f->T2C, st));
            SHERF_RUN(sherf_smpl_c2s_table(f->weights, A1, A2, PO1, SO2, PO2, f->obs_R, f->obs_Th, f->cam_R, f->cam_T, f->cam_K,
                                           f->C2S, st));
            SHERF_HIP_CHECK(hipEventRecord(d.ev_smpl, as_stream(st)));
            SHERF_PROF(1, as_stream(st));
            return SHERF_OK;
        };
        if (!stream_aux) SHERF_RUN(smpl_tables(stream_side));
        const int stagger = f->main_after_layer;
        const int half_tables = (f->flags & SHERF_FRAME_HALF_TABLES) ? 1 : 0;      // folded tables + voxel rows in fp16 (the image stays fp32)
        auto fold_tables = [&](sherf_stream_t st) -> int {      // per-frame table re-layout (channel-last, projections folded in)
            SHERF_RUN(sherf_fold_tables(f->planes, f->Wa_t, f->planes_f, f->P * f->P, 3, 32, (int64_t)f->P * f->P * 32, half_tables, st));
            SHERF_RUN(sherf_fold_tables(f->obs_feat, f->Wb_t, f->feat_f, f->Hf * f->Wf, 2, 64, 32, half_tables, st));
            return sherf_img_to_hwc4(f->obs_img, f->img4, f->H * f->W, st);
        };
        SHERF_CAP_TRACE("fold tables on aux");
        if (stream_aux) {       // independent of rays and voxels: off the ray side's chain, ahead of the level builds
            SHERF_HIP_CHECK(hipStreamWaitEvent(as_stream(stream_aux), d.ev_start, 0));
            SHERF_RUN(fold_tables(stream_aux));
            SHERF_HIP_CHECK(hipEventRecord(d.ev_fold, as_stream(stream_aux)));
        }
        const size_t ncell1 = (size_t)SHERF_MAX_CELLS + 1;
        auto enqueue_encoder = [&]() -> int {        // ---- side: a11 sparse voxel encoder ----
            const std::function<int()> smpl_on_aux = [&]() -> int { return smpl_tables(stream_aux); };
            SHERF_RUN(sherf_svox_encode_impl(f->vox_plan, f->vox_coord, f->vox_feat, f->vox_n, f->vox_training, levels, stream_side,
                                             stagger >= 0 ? d.ev_mid : nullptr, stagger, stream_aux, d.ev_lev,
                                             stream_aux ? &smpl_on_aux : nullptr,
                                             half_tables | ((f->flags & SHERF_FRAME_ENCODER_SINGLE) ? 2 : 0)));
            SHERF_HIP_CHECK(hipEventRecord(d.ev_enc, side));
            SHERF_PROF(2, side);
            return SHERF_OK;
        };
        // Launch order == start order (a launch costs the host ~5 us): staggered -> encoder first, the ray side waits for
        // layer `stagger`; concurrent -> the ray side's big kernels first, the encoder's ~50 small ones behind them.
        const bool encoder_first = stagger >= 0 || (xp & 8);
        SHERF_CAP_TRACE("cells");
        if (encoder_first) SHERF_RUN(enqueue_encoder());
        // ---- main: cell lists, a4-a6 sampling / mask / nearest vertex / compaction, table re-layout ----
        const bool lists = f->near_hdr && f->near_list;             // exact vertex list per near-mask sub-cell: its counts also give the mask
        SHERF_RUN(sherf_build_cells2(f->verts, f->Rg, f->Th, f->tverts, V, 0.05f, f->grid_hdr, f->cell_start, f->cell_pts,
                                     f->cell_scratch, lists ? nullptr : f->near_mask, stream_main));
        if (stagger >= 0) SHERF_HIP_CHECK(hipStreamWaitEvent(main, stagger < f->vox_plan->n_layers ? d.ev_mid : d.ev_enc, 0));
        SHERF_CAP_TRACE("near lists");
        if (lists)
            SHERF_RUN(sherf_build_near_lists(f->grid_hdr, f->cell_pts, V, 0.05f, f->near_hdr, f->near_list, f->near_list_cap, f->near_mask, stream_main));
        SHERF_CAP_TRACE("sampler");
        SHERF_RUN(sherf_sample_mask_nn_impl(f->ray_o, f->ray_d, f->near, f->far, f->R, f->S, f->Rg, f->Th, f->grid_hdr, f->cell_start,
                                       f->cell_pts, f->near_mask, f->capacity, f->counters, f->ray_base, f->ray_cnt, f->cs_idx,
                                       f->cs_vid, f->cs_xs, f->dense_vid, f->ray_mask, f->scan_ws, f->near_hdr, f->near_list, stream_main, f->sticky));
        SHERF_CAP_TRACE("sampler queued");
        // SHERF_FRAME_EXACT_GRIDS: the kernels after the compaction are launched for the frame's ACTUAL number of valid samples instead
        // of the buffers' capacity (R*S, of which a body fills a few percent: the MLP's grid is then ~96 % workgroups that allocate
        // 8 waves x 250 VGPRs + 85 KiB LDS only to read the count and exit, one at a time per CU, behind the real ones).  The count is
        // copied to pinned memory here and awaited just before the warp is enqueued -- the one host synchronisation of this mode (the
        // reference synchronises at the same point: its boolean-mask indexing, renderer.py:320-321); the encoder chain and the table
        // folds are enqueued in between and keep the GPU busy meanwhile.  Same results: every kernel clamps to min(count, capacity).
        if (f->flags & SHERF_FRAME_REPORT_COUNT) {
            const int slot = d.rep_next;
            d.rep_next = (d.rep_next + 1) % 8;
            SHERF_HIP_CHECK(hipMemcpyAsync(d.host_nv + 8 + slot, f->counters, sizeof(int32_t), hipMemcpyDeviceToHost, main));
            SHERF_HIP_CHECK(hipEventRecord(d.ev_rep[slot], main));
            t_rep_last[dev] = slot;
            d.rep_seq[slot] = t_rep_seq[dev] = ++d.rep_counter;
        }
        const bool exact = (f->flags & SHERF_FRAME_EXACT_GRIDS) != 0;
        if (exact) {
            SHERF_HIP_CHECK(hipMemcpyAsync(d.host_nv, f->counters, sizeof(int32_t), hipMemcpyDeviceToHost, main));
            SHERF_HIP_CHECK(hipEventRecord(d.ev_cnt, main));
        }
        SHERF_CAP_TRACE("encoder");
        if (!encoder_first) SHERF_RUN(enqueue_encoder());
        SHERF_CAP_TRACE("encoder queued");
        if (!stream_aux) SHERF_RUN(fold_tables(stream_main));
        // ---- main: a8-a10 warp, a10-a12 gather, a13-a14 MLP ----
        SHERF_HIP_CHECK(hipStreamWaitEvent(main, d.ev_smpl, 0));
        if (stream_aux) SHERF_HIP_CHECK(hipStreamWaitEvent(main, d.ev_fold, 0));
        int64_t cap = tok_cap;
        if (exact) {
            SHERF_HOST_STAMP(xp, "count wait begins");
            SHERF_HIP_CHECK(hipEventSynchronize(d.ev_cnt));
            SHERF_HOST_STAMP(xp, "count wait ends");
            int64_t c = *d.host_nv > 0 ? ((int64_t)*d.host_nv + 255) / 256 * 256 : 256;      // whole MLP tile groups
            if (c < cap) cap = c;
        }
        SHERF_CAP_TRACE("warp");
        SHERF_RUN(sherf_warp_geom(f->counters, f->cs_idx, f->cs_vid, f->cs_xs, f->ray_d, f->S, f->Rg, f->T2C, f->C2S, f->tverts,
                                  f->grid_hdr + kGridHdr, f->cell_start + ncell1, f->cell_pts + (size_t)V * 4, cap, f->geom,
                                  f->cs_tvid, stream_main));
        const int gv = ((f->gather_split & 2) ? 4 : 0) | ((f->gather_split & 4) ? 12 : 0) |  // bit 1: branchless voxel-row loads (mode | 4); bit 2: in 128 VGPRs (mode | 12)
                       (half_tables ? 16 : 0);                                                // mode | 16: fp16 tables
        // Gather and MLP in `nparts` contiguous parts of the tile list, part k's MLP (matrix pipe) on the side stream beside part k + 1's
        // gather (texture addresser) on the main one: the two kernels are bound by different units of the CU (frame->mlp_parts).
        const int nparts = std::min(((g_sherf_debug >> 16) & 15) ? ((g_sherf_debug >> 16) & 15) : f->mlp_parts, kMaxParts);   // (debug bits 16-19 override: A/B runs)
        const bool split_form = f->zfrag && (((f->flags & SHERF_FRAME_MLP_SPLIT) != 0) != ((g_sherf_debug & 8192) != 0));
        // SHERF_FRAME_PE_FRAGS (round 6): the gather writes the positional encodings as fp16 operand fragments and the pipelined single-fp16-product
        // network reads them (sherf_gather_tokens_pe -> sherf_nerf_mlp3_pe; bit-identical frames).  Only in the configuration it is built for: fp16
        // tables in the eight-channel gather, one pass, one part, the pipelined form; anything else renders as before.
        const bool pe_frags = (f->flags & SHERF_FRAME_PE_FRAGS) && f->pefrag && half_tables && (f->mlp_prec & 255) == 2 && (f->flags & SHERF_FRAME_MLP_PIPELINED) &&
                              !(g_sherf_debug & (1 << 25)) && !split_form && nparts <= 1 && !(f->gather_split & 7) && !(g_sherf_debug & (2048 | (1 << 29)));
        if (nparts > 1 && !(f->gather_split & 1) && !split_form) {
            SHERF_PROF(3, main);
            SHERF_HIP_CHECK(hipStreamWaitEvent(main, d.ev_enc, 0));
            for (int k = 0; k < nparts; ++k) {
                SHERF_RUN(sherf_gather_tokens(f->counters, f->geom, f->planes_f, f->P, f->feat_f, f->Hf, f->Wf, f->img4, f->H, f->W,
                                              levels, f->tok_bias, f->bounds, f->vox_min, f->vox_sh, 0 | gv | (k << 8) | (nparts << 16), cap,
                                              f->tokens, f->extras, stream_main));
                SHERF_HIP_CHECK(hipEventRecord(d.ev_part[k], main));
                SHERF_HIP_CHECK(hipStreamWaitEvent(side, d.ev_part[k], 0));
                if ((f->mlp_prec & 255) != 1 && (f->flags & SHERF_FRAME_MLP_PIPELINED))       // two workgroups per CU: room for the next part's gather beside it
                    SHERF_RUN(sherf_nerf_mlp3_part(f->counters, f->tokens, f->extras, f->wstream, f->wbias, f->mlp_prec, cap, f->sample_out, k, nparts,
                                                   (g_sherf_debug & (1 << 26)) ? 3 : (g_sherf_debug & (1 << 27)) ? 1 : 2, stream_side));   // (debug bits 26 / 27: residency 3 / 1, for A/B runs)
                else
                    SHERF_RUN(sherf_nerf_mlp_part(f->counters, f->tokens, f->extras, f->wstream, f->wbias, f->mlp_prec, cap, f->sample_out, k, nparts,
                                                  stream_side));
            }
            SHERF_PROF(4, main);
            SHERF_HIP_CHECK(hipEventRecord(d.ev_mlp, side));
            SHERF_HIP_CHECK(hipStreamWaitEvent(main, d.ev_mlp, 0));
            SHERF_PROF(5, main);
        } else {
        if (f->gather_split & 1) {      // tri-plane + pixel taps do not need the encoder: run them while it is still busy
            SHERF_RUN(sherf_gather_tokens(f->counters, f->geom, f->planes_f, f->P, f->feat_f, f->Hf, f->Wf, f->img4, f->H, f->W,
                                          nullptr, f->tok_bias, f->bounds, f->vox_min, f->vox_sh, 1 | gv, cap, f->tokens,
                                          f->extras, stream_main));
            SHERF_PROF(3, main);
            SHERF_HIP_CHECK(hipStreamWaitEvent(main, d.ev_enc, 0));
            SHERF_RUN(sherf_gather_tokens(f->counters, f->geom, f->planes_f, f->P, f->feat_f, f->Hf, f->Wf, f->img4, f->H, f->W,
                                          levels, f->tok_bias, f->bounds, f->vox_min, f->vox_sh, 2 | gv, cap, f->tokens,
                                          f->extras, stream_main));
        } else {
            SHERF_PROF(3, main);
            SHERF_HIP_CHECK(hipStreamWaitEvent(main, d.ev_enc, 0));
            if (pe_frags)
                SHERF_RUN(sherf_gather_tokens_pe(f->counters, f->geom, f->planes_f, f->P, f->feat_f, f->Hf, f->Wf, f->img4, f->H, f->W,
                                                 levels, f->tok_bias, f->bounds, f->vox_min, f->vox_sh, 0 | gv, cap, f->tokens,
                                                 f->extras, f->pefrag, stream_main));
            else
            SHERF_RUN(sherf_gather_tokens(f->counters, f->geom, f->planes_f, f->P, f->feat_f, f->Hf, f->Wf, f->img4, f->H, f->W,
                                          levels, f->tok_bias, f->bounds, f->vox_min, f->vox_sh, 0 | gv, cap, f->tokens,
                                          f->extras, stream_main));
        }
        SHERF_CAP_TRACE("network");
        SHERF_PROF(4, main);
        // debug bit 13 flips the form for A/B runs in one process (a frame without zfrag always takes the one-launch kernel)
        const bool split = f->zfrag && (((f->flags & SHERF_FRAME_MLP_SPLIT) != 0) != ((g_sherf_debug & 8192) != 0));
        if (pe_frags)
            SHERF_RUN(sherf_nerf_mlp3_pe(f->counters, f->tokens, f->extras, f->pefrag, f->wstream, f->wbias, f->mlp_prec, cap, f->sample_out, stream_main));
        else if (split)
            SHERF_RUN(sherf_nerf_mlp_split(f->counters, f->tokens, f->extras, f->wstream, f->wbias, f->mlp_prec, cap, f->zfrag,
                                           f->sample_out, stream_main));
        else if ((f->mlp_prec & 255) != 1 && (((f->flags & SHERF_FRAME_MLP_PIPELINED) != 0) != ((g_sherf_debug & (1 << 25)) != 0)))    // (debug bit 25 flips the form: A/B runs)
            SHERF_RUN(sherf_nerf_mlp3(f->counters, f->tokens, f->extras, f->wstream, f->wbias, f->mlp_prec, cap, f->sample_out, stream_main));
        else if ((f->mlp_prec & 255) != 1 && (((f->flags & SHERF_FRAME_MLP_TWO_TILES) != 0) != ((g_sherf_debug & (1 << 24)) != 0)))   // (debug bit 24 flips the form: A/B runs)
            SHERF_RUN(sherf_nerf_mlp2(f->counters, f->tokens, f->extras, f->wstream, f->wbias, f->mlp_prec, cap, f->sample_out, stream_main));
        else
            SHERF_RUN(sherf_nerf_mlp(f->counters, f->tokens, f->extras, f->wstream, f->wbias, f->mlp_prec, cap,
                                     f->sample_out, stream_main));
        SHERF_PROF(5, main);
        }
    }
    if (phase & 2) {
        // ---- a15-a16 ----
        SHERF_CAP_TRACE("compositing");
        SHERF_RUN(sherf_composite_compact_cap(f->counters, f->ray_base, f->ray_cnt, f->cs_idx, f->sample_out, f->ray_d, f->near, f->far,
                                              f->R, f->S, f->white_back, tok_cap, f->rgb, f->depth, f->acc, stream_main));
        SHERF_PROF(6, main);
        SHERF_HOST_STAMP(sherf_experiment(), "frame leave");
        if (cur_slot >= 0) {
            std::lock_guard<std::mutex> lk(g_mu);
            g_prof_host_ms[cur_slot] = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - host_t0).count();
            ++g_prof_n;
            cur_slot = -1;
        }
    }
#undef SHERF_PROF
    return SHERF_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// hipGraph replay of a frame (round 6).  A frame is ~65 launches, ~20 event operations and a memset on three streams:
// 0.2-0.28 ms of host time per call, at the head of every frame's latency.  Its launch sequence is a pure function of the descriptor -- every
// pointer, size, flag and the encoder plan it names -- the three stream handles, the debug / experiment words and the device: none of the
// data-dependent quantities (the valid-sample count, the voxel levels' row counts) ever reaches the host.  So the SECOND consecutive call with
// the same key is captured (hipStreamBeginCapture on the caller's stream; the side / aux streams join the capture through the library's own
// fork / join events, exactly as they join the frame) and instantiated, and from then on a frame costs one hipGraphLaunch.  A frame with new
// pointers (fresh input tensors, a re-sized workspace, other weights) has a new key: it is enqueued launch by launch as before and starts its own
// count, so sequences lose nothing.  Not captured: frames that report their count (SHERF_FRAME_REPORT_COUNT / _EXACT_GRIDS: a per-call pinned
// slot / a host wait), profiled frames (sherf_profile_frames: per-frame timing events), the host-stamp experiments, phase 2 / 4 alone.  A capture
// that fails for any reason renders the frame eagerly and never tries that key again.
// MEASURED on the MI355X (profiles/r06_call_h_*, cfg2_dense_ri, arms interleaved, bit-identical frames): replayed frames 1.719 ms against 1.722 ms
// enqueued launch by launch in the same two-stream form -- the GPU side gains NOTHING (the host runs ahead of a 1.7 ms frame either way; the
// dependent chains are bound by their kernels, not by launch gaps) -- and the three-stream form this runtime cannot capture (hipStreamEndCapture
// crashes, see `eligible` below) is 1.699 ms.  What a replay saves is the host's ~0.25 ms of enqueue work per frame.  So the mechanism is OPT-IN:
// environment SHERF_FRAME_GRAPH=1 or sherf_frame_graphs(1) (and a caller that passes no third stream); off by default.  sherf_frame_graph_stats
// reads what happened.
namespace {

struct GraphEntry {
    uint64_t key = 0, stamp = 0;
    int state = 0;                    // 0 free, 1 seen once (eager), 2 captured, 3 capture failed (eager for good)
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    sherf_vox_level lv[3];            // what the encoder's enqueue wrote into `levels_out_host` (host-side output of the call)
};
constexpr int kGraphSlots = 8;
GraphEntry g_graphs[kMaxDev][kGraphSlots];
uint64_t g_graph_clock = 0;
int g_graph_on = -1;                  // -1: read SHERF_FRAME_GRAPH at the first frame
bool g_graph_aux_ok = false;          // SHERF_FRAME_GRAPH_AUX=1: capture three-stream frames too (crashes the runtime on this box: see sherf_render_frame)
int64_t g_graph_stats[4] = {0, 0, 0, 0};   // captures, replays, eager frames, failed captures

inline void fnv(uint64_t& h, const void* p, size_t n) {
    const unsigned char* b = static_cast<const unsigned char*>(p);
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; }
}

void graph_drop(GraphEntry& e) {
    if (e.exec) (void)hipGraphExecDestroy(e.exec);
    if (e.graph) (void)hipGraphDestroy(e.graph);
    e = GraphEntry();
}

}  // namespace

extern "C" int sherf_frame_graphs(int enable) {
    std::lock_guard<std::mutex> frame_lock(g_frame_mu);
    g_graph_on = enable ? 1 : 0;
    if (!enable)
        for (auto& dev : g_graphs)
            for (auto& e : dev) graph_drop(e);
    return SHERF_OK;
}

extern "C" int sherf_frame_graph_stats(int64_t* stats_host, int32_t n) {
    SHERF_CHECK_ARG(stats_host && n >= 4);
    std::lock_guard<std::mutex> frame_lock(g_frame_mu);
    for (int i = 0; i < 4; ++i) stats_host[i] = g_graph_stats[i];
    return SHERF_OK;
}

extern "C" int sherf_render_frame(const sherf_frame* f, int phase, sherf_vox_level* levels, sherf_stream_t stream_main,
                                  sherf_stream_t stream_side, sherf_stream_t stream_aux) {
    SHERF_CHECK_ARG(f && levels && ((phase & 3) || phase == 4) && stream_side != stream_main && (!stream_aux || (stream_aux != stream_main && stream_aux != stream_side)));
    std::lock_guard<std::mutex> frame_lock(g_frame_mu);
    if (g_graph_on < 0) {
        const char* e = getenv("SHERF_FRAME_GRAPH");
        g_graph_on = (e && atoi(e) != 0) ? 1 : 0;                     // OFF unless asked for: see the note above sherf_render_frame
        const char* ea = getenv("SHERF_FRAME_GRAPH_AUX");
        g_graph_aux_ok = ea && atoi(ea) != 0;
    }
    int dev = 0;
    bool prof;
    { std::lock_guard<std::mutex> lk(g_mu); prof = g_prof_on; }
    const int xp = sherf_experiment();
    const bool eligible = g_graph_on == 1 && (phase == 1 || phase == 3) && !prof && !(f->flags & (SHERF_FRAME_EXACT_GRIDS | SHERF_FRAME_REPORT_COUNT)) &&
                          !(xp & (16 | 32)) && f->vox_plan && hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < kMaxDev && g_dev[dev].init &&
                          // MEASURED (MI355X, ROCm 7.0 runtime of PyTorch 2.10; profiles/r06_call_d_f_*): hipStreamEndCapture CRASHES on the frame's
                          // three-stream capture (main + side + aux with their cross dependencies) inside the product process, while the two-stream
                          // capture works (and the same three-stream shape works in a stand-alone probe, tools/ubench/graph_probe.hip).  Until
                          // that is understood only frames WITHOUT the third stream are captured (SHERF_FRAME_GRAPH_AUX=1 lifts the guard).
                          (!stream_aux || g_graph_aux_ok);
    if (!eligible) {
        ++g_graph_stats[2];
        return render_frame_enqueue(f, phase, levels, stream_main, stream_side, stream_aux);
    }
    uint64_t key = 1469598103934665603ull;
    fnv(key, f, sizeof(*f));
    fnv(key, f->vox_plan, sizeof(*f->vox_plan));
    const uint64_t words[7] = {(uint64_t)phase, (uint64_t)(size_t)stream_main, (uint64_t)(size_t)stream_side, (uint64_t)(size_t)stream_aux,
                               (uint64_t)(unsigned)g_sherf_debug, (uint64_t)(unsigned)xp, (uint64_t)dev};
    fnv(key, words, sizeof(words));
    if (key == 0) key = 1;
    GraphEntry* slot = nullptr;
    GraphEntry* lru = &g_graphs[dev][0];
    for (auto& e : g_graphs[dev]) {
        if (e.state && e.key == key) { slot = &e; break; }
        if (e.stamp < lru->stamp) lru = &e;
    }
    hipStream_t main = as_stream(stream_main);
    if (slot && slot->state == 2) {                                   // replay
        slot->stamp = ++g_graph_clock;
        for (int i = 0; i < 3; ++i) levels[i] = slot->lv[i];
        SHERF_HIP_CHECK(hipGraphLaunch(slot->exec, main));
        ++g_graph_stats[1];
        return SHERF_OK;
    }
    if (!slot) {                                                      // first sighting: eager, remember the key
        graph_drop(*lru);
        lru->key = key; lru->state = 1; lru->stamp = ++g_graph_clock;
        ++g_graph_stats[2];
        return render_frame_enqueue(f, phase, levels, stream_main, stream_side, stream_aux);
    }
    slot->stamp = ++g_graph_clock;
    if (slot->state == 3) {
        ++g_graph_stats[2];
        return render_frame_enqueue(f, phase, levels, stream_main, stream_side, stream_aux);
    }
    // second sighting: capture (on the library's own stream: the caller's may be the legacy default stream), instantiate, launch on the caller's
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    int rc = SHERF_ELAUNCH;
    const char* stage = "stream";
    hipError_t e = hipSuccess;
    DevState& d = g_dev[dev];
    if (!d.cap_stream) e = hipStreamCreateWithFlags(&d.cap_stream, hipStreamNonBlocking);
    bool ok = e == hipSuccess && d.cap_stream;
    g_sherf_cap_trace = getenv("SHERF_FRAME_GRAPH_DEBUG") ? 1 : 0;
    SHERF_CAP_TRACE("begin");
    if (ok) { stage = "begin"; e = hipStreamBeginCapture(d.cap_stream, hipStreamCaptureModeThreadLocal); ok = e == hipSuccess; }
    if (ok) {
        stage = "enqueue";
        rc = render_frame_enqueue(f, phase, levels, reinterpret_cast<sherf_stream_t>(d.cap_stream), stream_side, stream_aux);
        SHERF_CAP_TRACE("end capture");
        e = hipStreamEndCapture(d.cap_stream, &graph);                // (always ends the capture, also after a failed enqueue)
        SHERF_CAP_TRACE("capture ended");
        if (rc == SHERF_OK) stage = "end";
        ok = rc == SHERF_OK && e == hipSuccess && graph != nullptr;
    }
    if (ok) { stage = "instantiate"; SHERF_CAP_TRACE(stage); e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0); ok = e == hipSuccess && exec != nullptr; }
    if (ok) { stage = "launch"; SHERF_CAP_TRACE(stage); e = hipGraphLaunch(exec, main); ok = e == hipSuccess; }
    SHERF_CAP_TRACE("done");
    g_sherf_cap_trace = 0;
    if (!ok) {
        if (getenv("SHERF_FRAME_GRAPH_DEBUG"))
            fprintf(stderr, "[sherf] frame graph: capture failed at '%s' (rc %d, %s: %s); this descriptor renders launch by launch\n", stage, rc,
                    hipGetErrorName(e), rc != SHERF_OK ? g_sherf_err : hipGetErrorString(e));
        (void)hipGetLastError();
        if (exec) (void)hipGraphExecDestroy(exec);
        if (graph) (void)hipGraphDestroy(graph);
        slot->state = 3;
        ++g_graph_stats[3]; ++g_graph_stats[2];
        return render_frame_enqueue(f, phase, levels, stream_main, stream_side, stream_aux);
    }
    slot->graph = graph; slot->exec = exec; slot->state = 2;
    for (int i = 0; i < 3; ++i) slot->lv[i] = levels[i];
    ++g_graph_stats[0];
    return SHERF_OK;
}
