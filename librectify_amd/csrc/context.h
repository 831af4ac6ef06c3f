// lr_context: one device, one stream, one reusable workspace (see include/librectify_amd.h).
#pragma once
#include <atomic>
#include <functional>
#include <mutex>
#include <vector>

#include "common.h"
#include "vp_host.h"

struct lr_context {
    int device = 0;
    hipStream_t stream = nullptr;

    // geometry of the workspace / last frame
    size_t cap_pix = 0;
    int cap_tiles = 0;
    int w = 0, h = 0;

    // Host-buffer entry points (the reference's only kind, image.cpp:11-19): frames go through one of two device
    // slots on a copy stream of their own, so the upload of a lane's next frame overlaps the kernels of its current
    // one.  Pageable sources are first copied (in row bands, by `upload_threads` host threads) into a pinned staging
    // buffer per slot; sources that are already page-locked (lr_host_alloc, hipHostMalloc, hipHostRegister) are
    // DMA-copied where they lie.
    hipStream_t copy_stream = nullptr;
    float* d_img_slot[2] = {nullptr, nullptr};
    size_t cap_slot[2] = {0, 0};
    float* h_stage[2] = {nullptr, nullptr};
    size_t cap_stage[2] = {0, 0};
    hipEvent_t ev_up[2] = {};
    hipEvent_t ev_wait = nullptr;  // (blocking-sync flag) what a batch lane sleeps on
    std::vector<hipEvent_t> band_ev;  // single host frames: one event per 4 MB upload band (the filter follows the bands)
    void* crew = nullptr;             // StagingCrew of single host frames: threads kept from call to call
    int crew_helpers = 0;
    bool sleep_in_wait = false;
    // Batch calls on host frames: a ring of device frames (and, for pageable frames, of page-locked staging buffers)
    // that ONE uploader fills in frame order on the copy stream, as far ahead of the lanes as the ring allows
    // (find_groups_batch).
    std::vector<float*> ring_img;
    std::vector<float*> ring_stage;
    std::vector<hipEvent_t> ring_ev;
    size_t ring_cap_pix = 0, ring_stage_cap_pix = 0;
    // stage 1
    float* dx = nullptr;
    float* dy = nullptr;
    uint8_t* dmask = nullptr;
    uint64_t* cand = nullptr;
    uint32_t* cand_count = nullptr;
    uint32_t* tile_max = nullptr;
    // stage 2
    uint32_t* tile_pass = nullptr;
    uint32_t* tile_off = nullptr;      // (the fused seed selection keeps its workgroups' status words here: zeroed when allocated)
    uint32_t fit_tag = 0;              // ... and of the last component scan (kernels_fit.hip: component_offsets_kernel)
    int register_slow_calls = 0;       // batch calls still to go with the staging copy after pinning frames in place turned out slow (context.hip: find_groups_batch)
    uint32_t select_tag = 0;           // tag of the last seed selection on this context (kernels_seeds.hip: seed_select_kernel)
    float* maxmag = nullptr;
    uint64_t* keys_a = nullptr;
    uint64_t* keys_b = nullptr;
    uint32_t* d_counts = nullptr;  // [0] n_seeds, [1] n_comp, [2] n_px, [3..] flood scratch
    int32_t* seed_idx = nullptr;
    int32_t* seed_bin = nullptr;
    float* seed_thr = nullptr;
    int32_t* seed_size = nullptr;
    // stage 3
    uint32_t* label = nullptr;
    int32_t* queue = nullptr;
    lramd::FloodBuffers fb;
    hipStream_t flood_aux = nullptr;       // second stream of the flood: way-point seeds' team walks beside a round's exploration
    std::vector<hipEvent_t> flood_fork, flood_join;
    bool flood_aux_on = true;              // off for the lanes of a batch call (their frames overlap each other instead)
    size_t fb_cap_seeds = 0;
    // stage 4
    uint32_t* comp_rank = nullptr;
    uint32_t* comp_seed = nullptr;
    uint32_t* comp_off = nullptr;
    uint32_t* cursor = nullptr;
    uint32_t* px_a = nullptr;
    uint32_t* px_b = nullptr;
    float* scratch_w = nullptr;
    LineSegment* d_lines = nullptr;
    void* temp = nullptr;
    size_t temp_bytes = 0;
    uint32_t* comp_large = nullptr;  // components of more than 64 pixels (sorted by a workgroup each)
    lramd::HugeSort huge;                   // ... of more than 2^14: buckets (kernels_fit.hip: huge_count_kernel)
    uint32_t seed_cap = 0;           // capacity the seed sort runs with (the seed count is not known when it is enqueued)
    uint32_t seed_cap_once = 0;      // test hook: capacity of the next frame's seed sort
    int frame_laps = 0;              // laps the last frame took (1; 2 if the seed sort overflowed or the flood needed more rounds)
    lramd::FloodProgress flood_prog;
    int small_frames = 0;  // frames in a row of at most a quarter of the workspace's capacity (ctx_ensure_image_capacity gives it back after eight)
    // filter_lines + peeling on the device (kernels_groups.hip)
    size_t cap_glines = 0, cap_flines = 0;
    float* d_tables = nullptr;      // 3 pencil tables (all lines, two ping-pong round tables) x 8 arrays x cap_glines
    uint32_t* d_orig = nullptr;     // 3 x cap_glines
    float* d_inl = nullptr;         // 4 x cap_glines: (h, length) of a round's inliers beyond those staged in LDS
    LineSegment* d_flines = nullptr;  // filtered (then grouped) lines
    uint32_t* d_gctl = nullptr;     // peeling control block (kGc*)
    float* d_gnorm = nullptr;       // bounding-box centre and scale
    float* d_models = nullptr;      // refit model of each round
    uint8_t* h_res = nullptr;       // pinned: header (counts, control block, models) + the first res_lines_cap lines
    size_t res_lines_cap = 0;
    // grow-on-demand workspaces of the opt-in paths
    float* d_refine_table = nullptr;
    size_t cap_refine_table = 0;
    void* d_refine_edges = nullptr;
    size_t cap_refine_edges = 0;
    unsigned long long* d_cht_acc = nullptr;
    size_t cap_cht = 0;
    uint32_t* d_cht_idx = nullptr;   // lines a peeling round of the diamond-space estimator takes out of the accumulator
    uint32_t* h_cht_idx = nullptr;   // pinned mirror
    size_t cap_cht_idx = 0;
    uint32_t* d_cht_peak = nullptr;  // {cell, value lo, value hi, -, votes lo, votes hi}
    uint32_t* h_cht_peak = nullptr;  // pinned mirror
    int cht_d = 128;                 // accumulator size of estimator 3
    // RANSAC
    size_t cap_lines = 0;
    float* d_model = nullptr;  // 8 arrays of cap_lines
    float* h_model = nullptr;  // pinned mirror
    size_t cap_iter = 0;
    unsigned long long* d_best_slots = nullptr;  // kRansacBestSlots words: a scoring launch's best (score, iteration), cleared by its reader
    // PROSAC / Hough weights (opt-in estimator)
    int32_t* d_pairs = nullptr;   // 2 x ht_pairs
    int32_t* h_pairs = nullptr;
    size_t cap_pairs = 0;
    float* d_peak = nullptr;      // 3 floats
    float* d_weights = nullptr;   // cap_lines
    float* h_weights = nullptr;
    uint32_t* d_samples = nullptr;  // 2 buffers x 2 x cap_chunk
    uint32_t* h_samples = nullptr;
    uint32_t* d_hcounts = nullptr;  // 2 x cap_chunk
    uint32_t* h_hcounts = nullptr;
    size_t cap_chunk = 0, cap_wlines = 0;
    uint32_t* d_rec = nullptr;       // PROSAC: new-best iterations of a chunk ([0] = how many) ...
    uint32_t* h_rec = nullptr;
    uint8_t* d_recflags = nullptr;   // ... and a row of inlier flags for each
    uint8_t* h_recflags = nullptr;
    size_t cap_recflags = 0;         // (per chunk buffer; there are two of each)
    hipEvent_t prosac_ev[2] = {nullptr, nullptr};  // end of a chunk's work on the stream
    std::vector<int> prosac_imin;    // prosac.h's Imin(2, n) by n (constants of prosac.h:62-66 only)
    std::vector<lr_context*> workers;  // extra contexts (own stream + workspace) for frames in flight in batch calls
    std::vector<lr_context*> peers;    // one context per entry of the device list of the last multi-device batch call (each with its own lanes)
    int batch_streams = 5;
    int estimator = 0;            // 0 = RANSAC (reference default), 1 = PROSAC, 2 = DirectEstimator, 3 = diamond space (CHT)
    int prosac_T_N = -1;
    // pinned host scalars
    uint32_t* h_counts = nullptr;  // 8 words of counts; + 16: the flood's control block; + 72: the words the flood's rounds report in
    float* h_best = nullptr;       // [0] score, [1] iter (as int bits)

    // constants
    lramd::FilterConsts fconsts;
    lramd::BinTrig trig;
    float seed_keep_ratio = 0.f;

    // state of the last run
    uint32_t n_seeds = 0, n_comp = 0, n_px = 0;
    int flood_rounds = 0;
    int flood_rounds_hint = 10;  // rounds the next flood enqueues blindly
    int flood_rounds_last = 0;   // rounds the last flood needed (0: none yet)
    int flood_jit_sleep_us = 0;  // (lanes of a batch call, when they enqueue just in time at all: pause between looks)
    bool flood_giant_step = true;  // the lowest active seed's flood by the whole device when it outgrows the LDS tiers (kernels_flood.hip: kCtrlGiantStep); lr_set_flood_giant_step, LIBRECTIFY_FLOOD_GIANT_STEP=0
    bool flood_jit = true;       // single calls enqueue the flood's later rounds just in time (kernels_flood.hip: flood_enqueue); LIBRECTIFY_FLOOD_JIT=0
    uint64_t ransac_seed = 0;
    int ransac_iters = lramd::kRansacMaxIter;
    int flood_mode = 1;
    bool flood_logbig_off = false;  // lanes of a batch call keep no logs of second-tier walks (context.hip: find_groups_batch)
    bool flood_logbig_hint = true;  // did the last frame have walks in the second tier? (their logs need a launch of their own per round: kernels_flood.hip, flood_rewalk_kernel)
    bool flood_big_hint = true;  // did the last frame's walks outgrow the first storage tier? (none yet: assume so)
    bool flood_hold_hint = false;  // did the last frame hold its weakest seeds back?
    int flood_staged_streak = 0;     // frames in a row that started staged (every sixteenth starts without the hint)
    bool flood_staged_hint = false;  // was the last frame one of overlapping giants (kernels_flood.hip: kCtrlStaged)?  Then this one starts on its strongest quarter
    bool flood_calm_hint = false;      // the last frame's walks all stayed in the first storage tier (FloodBuffers::calm_hint)
    uint32_t flood_tiers[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // last flood: seeds in the second tier, slabs used, seeds of the ordered tail, hold-back, walked px (lo, hi), steps (lo, hi), walks beyond the first tier's table, multi-source re-walks, re-walks from logs, logs given up, giants held back
    bool flood_partial = true;  // partial commits of blocked seeds (kernels_flood.hip); lr_set_flood_partial_commits
    int flood_log_min = 0, flood_log_walk = 0;  // thresholds of the logs (0: the defaults; the lanes of a batch call get 32 and 24)
    int flood_log_from = 1;        // first round (from 0) whose seeds turn to their logs (lanes of a batch: experiment knob LIBRECTIFY_FLOOD_LOGS_LANES_FROM)
    bool flood_log_sweep = false;  // test hook (lr_set_flood_logs(ctx, 2)): every footprint worked out from a log goes the fall-back way (sweeps)
    bool flood_logs = true;     // blocked seeds work their next footprint out from the records of their last walk (kernels_flood.hip: flood_rewalk_kernel); lr_set_flood_logs, LIBRECTIFY_FLOOD_LOGS=0
    bool flood_multi = false;   // re-walks of long footprints from several way-points at once (kernels_flood.hip): opt-in, lr_set_flood_multi_source / LIBRECTIFY_FLOOD_MULTI=1
    bool flood_staged = false;  // lr_set_flood_staged: the rounds start on the strongest eighth of the seeds (test / experiment hook)
    // Stage timers (HIP events between the stages of a frame): off in the frame calls unless lr_set_stage_timing or
    // LIBRECTIFY_STAGE_TIMES asks -- every event record is a barrier packet in the stream, some 6 us of idle GPU each,
    // seven of them per frame.  The staged API (lr_stage_*) always times its stages.
    bool timing_on = false;
    hipEvent_t ev[16] = {};
    float stage_ms[LR_T_COUNT] = {};
    double host_ms[3] = {0, 0, 0};  // last frame: enqueue, next-frame staging + upload, wait (LIBRECTIFY_LANE_DEBUG)
    bool stage_valid[4] = {false, false, false, false};
    bool dmask_consumed = false;  // the parallel flood clears the mask of labelled pixels: LR_BUF_DMASK is then stale
};

namespace lramd {
constexpr size_t kResHeaderBytes = 256;
int ctx_create(int device, lr_context** out);
void ctx_destroy(lr_context* c);
const std::string& get_error();
int ctx_ensure_image_capacity(lr_context* c, int w, int h);
int ctx_trim(lr_context* c, bool frames_too);
int ctx_ensure_ransac_capacity(lr_context* c, size_t n_lines, size_t n_iter);
int ctx_stage_filter(lr_context* c, const float* d_image, int w, int h, int stride);
int ctx_stage_seeds(lr_context* c);
int ctx_stage_flood(lr_context* c);
int ctx_stage_fit(lr_context* c, std::vector<LineSegment>& out);
int ctx_detect(lr_context* c, const float* d_image, int w, int h, int stride, std::vector<LineSegment>& raw);
int ctx_ransac_best(lr_context* c, const PencilModel& model, const std::vector<int>& indices, float tol, int n_iter,
                    uint64_t seed, uint32_t round, Vec3* best_h, float* best_score, int* best_iter);
struct ProsacTrace {
    int iterations = 0, n_star = 0, best_iter = -1, I_N_best = 0;
};
int ctx_ht_weights(lr_context* c, const PencilModel& model, const std::vector<int>& indices, std::vector<float>& weights);
int ctx_prosac_solve(lr_context* c, const PencilModel& model, const std::vector<int>& indices, float tol, int T_N,
                     uint64_t seed, uint32_t round, Vec3* h, ProsacTrace* trace);
int ctx_estimate_line_pencils_prosac(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                                     float garbage_deg, int T_N, uint64_t seed);
int ctx_estimate_line_pencils(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                              float garbage_deg, int n_iter, uint64_t seed);
int ctx_direct_solve(lr_context* c, const PencilModel& model, const std::vector<int>& indices, Vec3* h);
int ctx_estimate_line_pencils_direct(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                                     float garbage_deg);
int ctx_cht_vanishing_point(lr_context* c, const std::vector<LineSegment>& lines, int d, Vec3* vp,
                            std::vector<uint64_t>* acc_out);
struct ChtTrace {
    std::vector<Vec3> models;          // refit of each round (normalised coordinates)
    std::vector<uint32_t> peak_cell;   // winning accumulator cell of each round
    uint64_t votes = 0;                // cells voted for (added or taken back) over the call
};
int ctx_estimate_line_pencils_cht(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                                  float garbage_deg, int d, ChtTrace* trace);
int ctx_refine(lr_context* c, std::vector<LineSegment>& lines);
int ctx_find_groups_device(lr_context* c, const float* d_image, int w, int h, int stride, float min_length, bool refine,
                           std::vector<LineSegment>& out);
int ctx_find_groups_host(lr_context* c, const float* buffer, int w, int h, int stride, float min_length, bool refine,
                         int num_threads, std::vector<LineSegment>& out);
int ctx_find_groups_batch_device(lr_context* c, const float* d_images, size_t image_stride, int batch, int w, int h,
                                 int stride, float min_length, bool refine, LineSegment* out, int capacity, int* n_lines,
                                 const RectificationConfig* cfg, ImageTransform* transforms);
// host-resident frames (any stride sign, pageable or page-locked): staged uploads overlap the kernels
int ctx_find_groups_batch_host(lr_context* c, const float* const* frames, int batch, int w, int h, int stride,
                               float min_length, bool refine, int num_threads, LineSegment* out, int capacity,
                               int* n_lines, const RectificationConfig* cfg, ImageTransform* transforms);
// The same over several devices of this process ("one host thread + stream set per device"): frames are
// dealt in contiguous blocks of ceil(batch / n_devices), block i to devices[i] (a device may be listed more than once:
// every entry gets a lane set of its own), results land in the caller's arrays; no collective, it is one process.
int ctx_find_groups_batch_host_multi(lr_context* c, const int* devices, int n_devices, const float* const* frames, int batch,
                                     int w, int h, int stride, float min_length, bool refine, int num_threads,
                                     LineSegment* out, int capacity, int* n_lines, const RectificationConfig* cfg,
                                     ImageTransform* transforms);
// Enqueues the upload of a host frame into device slot `slot` on the copy stream and records ev_up[slot];
// the caller makes its compute stream wait on that event.  num_threads: the reference's knob (threading.h:24-27),
// here the number of host threads that stage a pageable frame (< 0: serial, as there).
int ctx_upload_frame(lr_context* c, int slot, const float* buffer, int w, int h, int stride, int num_threads);
}  // namespace lramd
