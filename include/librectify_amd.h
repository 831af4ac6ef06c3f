/*
 * librectify_amd.h — extensions of the MI355X build beyond the reference's six functions.
 *
 * Plain C ABI (pointers and sizes only).  Everything here is additive: the drop-in symbols
 * of include/librectify.h behave as the reference's (src/interface.cpp) and are thin
 * wrappers over a thread-local context of this API.
 *
 * Why these exist:
 *  - the reference API takes a host buffer and has no device/stream notion
 *    (src/librectify.h:111-116); a caller that already holds frames in HBM, or wants many
 *    frames in flight, needs device-pointer and batch entry points;
 *  - parity tests and bench.py need stage-level access (filter / seeds / flood / fit /
 *    RANSAC scoring) and per-stage HIP-event timings.
 *
 * All functions return 0 on success, non-zero on failure (lr_last_error() has the text).
 * No function falls back to a CPU path: if no GPU is present they fail.
 */
#pragma once
#include <stddef.h>
#include <stdint.h>

#include "librectify.h"

#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#ifdef __cplusplus
using librectify::ImageTransform;
using librectify::LineSegment;
using librectify::Point;
using librectify::RectificationConfig;
extern "C" {
#endif

typedef struct lr_context lr_context;

/* ---- context ------------------------------------------------------------------------- */
/* One context = one device, one HIP stream, one reusable workspace.  Not thread-safe; use
 * one context per host thread (the drop-in functions do exactly that). */
int lr_context_create(int device, lr_context** out);
void lr_context_destroy(lr_context* ctx);
/* The drop-in functions keep one context per calling host thread (device workspace of about 130 bytes per pixel of the
 * largest frame seen, 288 MB of flood overflow slabs, 32 MB of hand-over records, page-locked staging, staging threads) until the thread exits;
 * a thread that is done with the library for a while can give it back at once.  The next call makes a new one. */
void lr_release_thread_context(void);
/* The reference is stateless; a context is not: its device workspace is sized by the LARGEST frame it has seen (an 8192 x 8192
 * call leaves 8.9 GB behind).  It shrinks by itself -- after eight frames in a row of at most a quarter of its capacity the
 * workspace is given back and allocated again at the size in use -- and on request: lr_context_trim frees everything that is
 * sized by frames (workspace, flood buffers and slabs, frame slots, page-locked staging, the batch ring; streams and events
 * stay), the next call allocates what it needs.  lr_trim_thread_context does that for the calling thread's drop-in context
 * and keeps the context, where lr_release_thread_context destroys it. */
int lr_context_trim(lr_context* ctx);
int lr_trim_thread_context(void);
const char* lr_last_error(void);
int lr_synchronize(lr_context* ctx);
/* RANSAC sample stream seed (the reference seeds from std::random_device, estimator.h:35;
 * this build is reproducible: default 0 or env LIBRECTIFY_SEED). */
void lr_set_ransac_seed(lr_context* ctx, uint64_t seed);
/* RANSAC iterations per model (reference config.h:36 RANSAC_MAX_ITER = 10000). */
void lr_set_ransac_iterations(lr_context* ctx, int n_iter);
/* Flood implementation: 0 = ordered single-wave (simple, slow), 1 = parallel rounds (default);
 * 2 / 3 / 4 = test hooks: parallel rounds without the second LDS storage tier and with no / two / all overflow
 * slabs (exhausted-storage and slab paths; lr_stage_counters tells which storage a frame used); 5 = second tier with
 * room for one seed per round and no slab (a stall while the weakest seeds are held back); 6 / 7 = the second tier's team of
 * wavefronts runs out of storage after 200 tiles and hands the walk to a slab / has no slab to hand it to.
 * All modes give identical results. */
void lr_set_flood_mode(lr_context* ctx, int mode);
int lr_device_count(void);

/* ---- full path ------------------------------------------------------------------------ */
/* find_line_segment_groups on an image already resident in HBM (row-major float, `stride`
 * elements between rows, stride >= width).  Writes at most `capacity` segments to the HOST
 * array `out`; *n_lines is the number found (0 on the reference's NULL paths,
 * interface.cpp:50-54,65-69). */
int lr_find_line_segment_groups_device(lr_context* ctx, const float* d_image, int width, int height, int stride,
                                       float min_length, int refine, int num_threads, LineSegment* out, int capacity,
                                       int* n_lines);
/* Same, host buffer (any stride sign, as the reference: image.cpp:11-19). */
int lr_find_line_segment_groups_host(lr_context* ctx, const float* buffer, int width, int height, int stride,
                                     float min_length, int refine, int num_threads, LineSegment* out, int capacity,
                                     int* n_lines);
/* Batch of `batch` device-resident frames of one size, frame b at d_images + b*image_stride.
 * Output b goes to out + b*capacity; n_lines[b]; transforms[b] (may be NULL) is
 * compute_rectification_transform(lines_b, cfg). */
int lr_find_line_segment_groups_batch_device(lr_context* ctx, const float* d_images, size_t image_stride, int batch,
                                             int width, int height, int stride, float min_length, int refine,
                                             int num_threads, LineSegment* out, int capacity, int* n_lines,
                                             const RectificationConfig* cfg, ImageTransform* transforms);

/* Batch of `batch` HOST-resident frames of one size, frame b at frames + b*image_stride (elements), rows `stride`
 * elements apart (any sign, as image.cpp:11-19).  This is the reference's own kind of input, many frames at once.
 * ONE uploader thread sends the frames, in frame order, on the context's (high-priority) copy stream into a pool of
 * lanes + 6 device frames (LIBRECTIFY_RING_EXTRA overrides the 6; never more than 2 GiB or `batch` frames: 0.5 GiB
 * of device memory for 4K frames with six lanes), so the link runs ahead of the lanes; a lane only makes its stream
 * wait for its frame's transfer.  Page-locked memory (lr_host_alloc, hipHostMalloc, hipHostRegister) is DMA-copied
 * where it lies.  Pageable memory goes through page-locked staging buffers -- as many as there are pool slots, i.e.
 * the same amount again in pinned host memory, allocated when a call first meets a pageable frame -- filled in 4 MB
 * row bands by `num_threads` host threads shared by the whole call (the reference's knob, threading.h:24-27: < 0 or
 * 1 = the uploader alone; capped at 8 and at the host's cores), started once per call.  Outputs as for the device
 * batch. */
int lr_find_line_segment_groups_batch_host(lr_context* ctx, const float* frames, size_t image_stride, int batch,
                                           int width, int height, int stride, float min_length, int refine,
                                           int num_threads, LineSegment* out, int capacity, int* n_lines,
                                           const RectificationConfig* cfg, ImageTransform* transforms);
/* Same with one pointer per frame. */
int lr_find_line_segment_groups_batch_host_ptrs(lr_context* ctx, const float* const* frames, int batch, int width,
                                                int height, int stride, float min_length, int refine, int num_threads,
                                                LineSegment* out, int capacity, int* n_lines,
                                                const RectificationConfig* cfg, ImageTransform* transforms);
/* The same batch over SEVERAL devices of this process -- what a C / C++ caller of the reference's kind (one process, no
 * launcher: src/autorectify.cpp:136,350) needs to use the eight GPUs of a node.  Frames are dealt in contiguous blocks of
 * ceil(batch / n_devices), block i to devices[i]; every entry of the list gets a context of its own (kept with `ctx` from
 * call to call, configured like `ctx`: seed, iterations, estimator, lr_set_batch_streams lanes per device) with its own
 * uploader, copy stream, lanes and pool, driven by a host thread of its own; results land in the caller's arrays as for
 * the single-device call.  A device may be listed more than once (two independent lane sets on one GPU: how the call is
 * tested on a one-GPU box).  No collective: it is one process.  `ctx` itself only carries the settings. */
int lr_find_line_segment_groups_batch_host_multi(lr_context* ctx, const int* devices, int n_devices,
                                                 const float* const* frames, int batch, int width, int height, int stride,
                                                 float min_length, int refine, int num_threads, LineSegment* out,
                                                 int capacity, int* n_lines, const RectificationConfig* cfg,
                                                 ImageTransform* transforms);
/* Page-locked host memory for frames (hipHostMalloc / hipHostFree): uploads from it skip the staging copy. */
int lr_host_alloc(lr_context* ctx, size_t bytes, void** out);
int lr_host_free(lr_context* ctx, void* p);

/* Minimal device-memory helpers (hipMalloc / hipFree / synchronous hipMemcpy H2D on the context's device). */
int lr_device_malloc(lr_context* ctx, size_t bytes, void** out);
int lr_device_free(lr_context* ctx, void* p);
int lr_memcpy_h2d(lr_context* ctx, void* dst, const void* src, size_t bytes);
/* Frames kept in flight by the batch call (one host thread + HIP stream + workspace each; default 5). */
void lr_set_batch_streams(lr_context* ctx, int n);
/* Test hooks for the two situations in which a frame takes a second lap (lr_stage_counters [7] tells): the capacity
 * the NEXT frame's seed sort starts with (normally 1.5 x the previous frame's seed count; a frame with more seeds is
 * repeated with room), and the number of flood rounds the next frame enqueues before it looks at the flood's control
 * block (normally the previous frame's rounds + 2; a flood that needs more is completed after the frame's wait, and
 * the stages after it run again). */
void lr_set_seed_capacity(lr_context* ctx, uint32_t cap);
/* Test / experiment hook: start the flood's rounds on the strongest eighth of the seeds and widen the window round by
 * round (same result; it was the batch lanes' setting in round 1). */
void lr_set_flood_staged(lr_context* ctx, int on);
void lr_set_flood_blind_rounds(lr_context* ctx, int rounds);
/* Comparison hook: the flood's partial commits (a blocked seed commits at once the part of its footprint that no lower
 * seed can reach; on by default, LIBRECTIFY_FLOOD_PARTIAL=0 also switches them off).  Same labels either way. */
void lr_set_flood_partial_commits(lr_context* ctx, int on);
/* Opt-in (round 4): the flood's multi-source re-walks.  A seed whose walk covered a hundred tiles or more leaves way-points
 * on its footprint; if it has to walk again, a team of wavefronts starts from the seed and from every way-point at once,
 * beside the round's exploration on a second stream, and keeps what is connected to the seed.  Same labels either way
 * (exact by construction, tested against the oracle); on the bench frames the late rounds get 25 % shorter and the whole
 * flood 2-10 %, frames of regions and of very long bars lose as much, hence off by default.
 * LIBRECTIFY_FLOOD_MULTI=1 turns it on for every new context; lr_stage_counters [10] counts such walks. */
void lr_set_flood_multi_source(lr_context* ctx, int on);
/* The flood's later rounds from the logs (kernels_flood.hip: flood_rewalk_kernel): a walk of twelve tiles or more leaves
 * its footprint as (tile, pixels) records, and since a footprint only ever shrinks (filter.cpp:101-153 accepts a pixel on
 * static data and on "not claimed yet"), the seed's next footprint is the connected part around it of those records minus
 * what has been committed since -- labelled in LDS instead of walked tile after tile.  Same labels.  On by default
 * (single 4K frames: flood 1.31 -> 0.92 ms); the lanes of a batch call keep logs only of walks of 32 tiles and more.
 * 0 = off, 1 = on, 2 = on, every log through the fall-back path (test hook).
 * LIBRECTIFY_FLOOD_LOGS=0 turns it off for every new context; lr_stage_counters [11], [12] count the logs worked on and
 * those that took the fall-back path. */
void lr_set_flood_logs(lr_context* ctx, int on);
/* Single calls enqueue the flood's first rounds blindly (what the context's last frame needed, less one; four at most) and every further
 * round only when the host has seen -- in page-locked words the last workgroup of a round writes -- that seeds are left: no
 * launch behind the last round with work (the blind rounds of a 4K frame were 60-120 us of empty launches); the calling
 * thread polls while the flood runs (a single call spins; a lane of a batch call looks every 20 us).  On by default.
 * 0 = blind rounds (the previous frame's count plus two: what lr_set_flood_blind_rounds steers), and a frame whose flood
 * needs more takes a second lap (lr_stage_counters [7]).  LIBRECTIFY_FLOOD_JIT=0 likewise. */
void lr_set_flood_just_in_time(lr_context* ctx, int on);
/* The giant step (kernels_flood.hip: kCtrlGiantStep).  The lowest active seed's flood is what the reference's loop
 * (line_detector.cpp:98-119 over filter.cpp:110-153) does next and nothing about it is speculative: when its walk outgrows
 * the LDS tiers (a smooth region of 100 000 pixels, a ring of a noiseless gradient) the whole device labels it between two
 * rounds -- the seed's acceptance test as a 64-bit mask per 8x8 tile, a union-find over the tiles' components, labels --
 * instead of one team of wavefronts walking it tile after tile through a global slab.  Same labels.  On by default;
 * 0 = the slab walk (comparison), LIBRECTIFY_FLOOD_GIANT_STEP=0 likewise; lr_stage_counters [14] counts the steps. */
void lr_set_flood_giant_step(lr_context* ctx, int on);

/* ---- stage API (tests, bench) --------------------------------------------------------- */
/* Stage 1: fused 5x5 derivative filter + magnitude + direction bin + dilated-bin mask +
 * 5x5 non-max candidates (reference line_detector.cpp:41-49,126-182, filter.cpp:29-98,161-168). */
int lr_stage_filter(lr_context* ctx, const float* d_image, int width, int height, int stride);
/* Same on a host buffer (uploaded to the context's staging image first; any stride sign). */
int lr_stage_filter_host(lr_context* ctx, const float* buffer, int width, int height, int stride);
/* Stage 2: global max, seed threshold, ordered seed list (line_detector.cpp:209-220, filter.cpp:168-194). */
int lr_stage_seeds(lr_context* ctx, int* n_seeds);
/* Stage 3: ordered flood (filter.cpp:101-153, line_detector.cpp:92-122).  The parallel flood (mode >= 1) CONSUMES the
 * filter output: it clears the direction mask of every pixel it labels, so LR_BUF_DMASK can no longer be downloaded
 * and a second flood of the same frame needs lr_stage_filter + lr_stage_seeds again (both fail with a message). */
int lr_stage_flood(lr_context* ctx, int* n_components);
/* Stage 4: weighted-PCA line fit per component (geometry.cpp:20-61); output in seed order. */
int lr_stage_fit(lr_context* ctx, LineSegment* out, int capacity, int* n_lines);

enum lr_buffer_id {
    LR_BUF_DX = 0,        /* float  w*h */
    LR_BUF_DY = 1,        /* float  w*h */
    LR_BUF_DMASK = 2,     /* uint8  w*h: bit b = pixel is inside the 3x3 dilation of grad_bin==b */
    LR_BUF_LABEL = 3,     /* int32  w*h: claiming seed index or -1 */
    LR_BUF_SEED_IDX = 4,  /* int32  n_seeds: row*w+col, canonical order */
    LR_BUF_SEED_BIN = 5,  /* int32  n_seeds */
    LR_BUF_SEED_THR = 6,  /* float  n_seeds: (1-TRACE_TOLERANCE)*value at the seed */
    LR_BUF_MAXMAG = 7,    /* float  1 */
    LR_BUF_SEED_SIZE = 8, /* int32  n_seeds: pixels claimed by each seed's flood (0 = skipped) */
};
int lr_download(lr_context* ctx, int buffer_id, void* dst, size_t bytes);

enum lr_stage_id {
    LR_T_UPLOAD = 0,
    LR_T_FILTER = 1,
    LR_T_SEEDS = 2,
    LR_T_FLOOD = 3,
    LR_T_FIT = 4,
    LR_T_RANSAC = 5,
    LR_T_TOTAL = 6,
    LR_T_FILTER_KERNEL = 7, /* the fused filter kernel alone */
    LR_T_COUNT = 8,
};
/* HIP-event times (ms) of the stages of the last call on this context. */
int lr_stage_times(lr_context* ctx, float* ms, int count);
/* The frame calls (lr_find_line_segment_groups_*, the batch entries and the drop-in symbol) record their stage timers
 * only when asked: lr_set_stage_timing(ctx, 1) or LIBRECTIFY_STAGE_TIMES in the environment (seven event records per
 * frame cost some 40 us of idle GPU).  lr_stage_times then returns the last frame's stages; the staged API below always
 * times its stages. */
void lr_set_stage_timing(lr_context* ctx, int on);
/* Duration (ms) of the last fused filter kernel alone (HIP events around its launch); valid right after
 * lr_stage_filter*, without running the later stages. */
int lr_filter_kernel_ms(lr_context* ctx, float* ms);
/* Extra counters of the last call: [0] seeds, [1] components, [2] flood rounds, [3] labelled pixels, and how the
 * flood's walks were stored: [4] seeds that moved to the second LDS tier, [5] global slabs used, [6] seeds finished by
 * the ordered single-wave tail (storage exhausted); [7] laps of the frame through the pipeline (1 normally); [8] pixels
 * the flood's explorations walked in all rounds together (over [3]: the re-walk factor), [9] their 8x8-tile steps, [10] re-walks that
 * started from several way-points at once (lr_set_flood_multi_source), [11] footprints worked out from a log instead of
 * walked (lr_set_flood_logs), [12] those of them that took the fall-back path (sweeps), [13] walks that outgrew the second
 * tier's table and were held back until their seed was the lowest active one (instead of moving into a global slab),
 * [14] giant steps: floods of the lowest active seed labelled by the whole device (lr_set_flood_giant_step), [15] walks that
 * outgrew the first tier in a round enqueued without the second (the context's last frame never needed it: such a walk waits
 * a round; LIBRECTIFY_FLOOD_CALM_HINT=0 enqueues every round with the second tier). */
int lr_stage_counters(lr_context* ctx, int64_t* out, int count);

/* ---- RANSAC --------------------------------------------------------------------------- */
/* Scores n_iter two-line hypotheses of the line-pencil model (line_pencil.cpp:89-140,
 * estimator.h:37-71) over the lines listed in `indices`, on the GPU.  lines_norm are
 * bbox-normalised segments (HOST).  Outputs the raw best hypothesis (first strictly best),
 * its score and iteration (-1 if every sample was degenerate). */
int lr_ransac_best(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float tol,
                   int n_iter, uint64_t seed, uint32_t round, float* best_h3, float* best_score, int* best_iter);
/* estimate_line_pencils (line_pencil.cpp:148-177): writes group_id in place (HOST array). */
int lr_estimate_line_pencils(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                             float garbage_deg, int n_iter, uint64_t seed);

/* Diamond-space ("cascaded Hough") accumulator, opt-in: what reference cht.h:13-24 describes (its cht.cpp is an
 * uncompiled sketch).  d x d accumulator (8 <= d <= 128) of length-weighted line polylines, kept in LDS with integer
 * atomics; returns the de-normalised vanishing point of the strongest pencil (z = 0: ideal point) and, if acc_out is
 * not NULL, the d*d 64-bit accumulator. */
int lr_cht_vanishing_point(lr_context* ctx, const LineSegment* lines, int n, int d, Point* vp, uint64_t* acc_out);
/* postprocess_lines_segments (line_detector.cpp:332-444), what refine=true runs: merges collinear neighbours.
 * `out` must hold n records; the O(n^2) pair test runs on the GPU for n >= 2048. */
int lr_refine_lines(lr_context* ctx, const LineSegment* in, int n, LineSegment* out, int* n_out);

/* ---- PROSAC / Hough weights (opt-in) --------------------------------------------------- */
/* The reference compiles prosac.h and DirectEstimator (estimator.h:82-96) but never instantiates them (ChangeLog.md:
 * "pure RANSAC is used"), so RANSAC is the default here too.  kind: 0 = RANSAC, 1 = PROSAC with T_N iterations
 * (<= 0: the reference's niter_RANSAC(0.9, 0.5, 2) = 9, prosac.h:116), 2 = DirectEstimator (refit on the lines whose
 * Hough weight exceeds 0.95; the parameter is ignored), 3 = the diamond-space accumulator (cht.h:13-24) with a
 * param x param accumulator (<= 0: 128; 8..128), see lr_estimate_line_pencils_cht. */
void lr_set_estimator(lr_context* ctx, int kind, int param);
/* estimate_multiple_structures (estimator.h:99-145) around the diamond-space accumulator as cht.h:13-24 describes
 * it: all lines vote once (length-weighted polylines, integer votes in LDS), every round takes the accumulator's
 * strongest cell as the hypothesis, the remaining lines within `inlier_deg` of it decide the refit (fit_optimal), the
 * refit's inliers get the round's id, near misses (< garbage_deg) are dropped, and the votes of both are taken back
 * out of the accumulator ("the weights can be negative (so lines can be removed!)", cht.h:18).  Writes group_id in
 * place (HOST array).  Optional outputs: models3 = refit of each round (3 floats, normalised coordinates),
 * n_models = rounds run, peak_cells = winning cell (row * d + column) of each round, votes = accumulator cells voted
 * for (added or taken back) during the call.  Parity unpinned: the reference's cht.cpp is an uncompilable sketch. */
int lr_estimate_line_pencils_cht(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                                 float garbage_deg, int d, float* models3, int* n_models, uint32_t* peak_cells,
                                 uint64_t* votes);
/* LinePencilModel::get_weights (line_pencil.cpp:47-86): 65x65 hemisphere accumulator (LDS, 64-bit integer
 * atomics), peak direction, inclination^4 per line listed in `indices`. */
int lr_ht_weights(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* weights);
/* PROSAC_Estimator::solve (prosac.h:104-299); trace4 = iterations run, n_star, iteration of the best sample, its inlier count. */
int lr_prosac_solve(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float tol,
                    int T_N, uint64_t seed, uint32_t round, float* h3, int32_t* trace4);
int lr_estimate_line_pencils_prosac(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                                    float garbage_deg, int T_N, uint64_t seed);
/* DirectEstimator::solve (estimator.h:82-96) and estimate_multiple_structures around it. */
int lr_direct_solve(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* h3);
int lr_estimate_line_pencils_direct(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                                    float garbage_deg);

#ifdef __cplusplus
}
#endif

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
