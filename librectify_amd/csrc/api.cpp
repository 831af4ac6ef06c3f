// C ABI of librectify_amd.so: the reference's six functions (include/librectify.h) and the
// lr_* extensions (include/librectify_amd.h).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <string>

#include "context.h"

using namespace lramd;

namespace {

struct CtxDeleter {
    void operator()(lr_context* c) const { ctx_destroy(c); }
};

// The reference API is stateless and re-entrant: each host thread lazily gets
// its own context on the device named by LIBRECTIFY_DEVICE (default 0).
std::unique_ptr<lr_context, CtxDeleter>& thread_context_slot() {
    thread_local std::unique_ptr<lr_context, CtxDeleter> ctx;
    return ctx;
}

lr_context* thread_context() {
    std::unique_ptr<lr_context, CtxDeleter>& ctx = thread_context_slot();
    if (!ctx) {
        const char* env = std::getenv("LIBRECTIFY_DEVICE");
        lr_context* c = nullptr;
        if (ctx_create(env ? std::atoi(env) : 0, &c)) return nullptr;
        ctx.reset(c);
    }
    return ctx.get();
}

// image_from_buffer (reference image.cpp:11-19): the frame goes to device slot 0 on the context's copy stream
// (pinned staging for pageable sources, see ctx_upload_frame) and the compute stream waits for it.
int upload_host_image(lr_context* c, const float* buffer, int width, int height, int stride, int num_threads) {
    if (ctx_upload_frame(c, 0, buffer, width, height, stride, num_threads)) return 1;
    LR_HIP(hipStreamWaitEvent(c->stream, c->ev_up[0], 0));
    return 0;
}

// No exception leaves the C ABI (the reference's convention: no exceptions, NULL / 0 for "nothing"):
// an allocation that fails inside an entry point becomes its error return, with the message in lr_last_error().
int guard_fail(const char* where) {
    try {
        throw;
    } catch (const std::bad_alloc&) {
        set_error(std::string(where) + ": out of host memory");
    } catch (const std::exception& e) {
        set_error(std::string(where) + ": " + e.what());
    } catch (...) {
        set_error(std::string(where) + ": unknown exception");
    }
    return 1;
}

int copy_out(const std::vector<LineSegment>& v, LineSegment* out, int capacity, int* n_lines) {
    const int n = (int)v.size();
    if (n_lines) *n_lines = n;
    if (out && capacity > 0) std::memcpy(out, v.data(), sizeof(LineSegment) * (size_t)std::min(n, capacity));
    return 0;
}

}  // namespace

extern "C" {

// ---- the reference's six entry points ----------------------------------------------------

LineSegment* find_line_segment_groups(float* buffer, int width, int height, int stride, float min_length, bool refine,
                                      int num_threads, int* n_lines) {
    // num_threads is the reference's OpenMP knob (threading.h:24-27).  The per-pixel stages run on the GPU
    // regardless; here it sets how many host threads stage a pageable frame for its upload (< 0: serial).
    if (n_lines) *n_lines = 0;
    set_error("");
    // The reference signals "nothing found" by NULL (interface.cpp:50-54,65-69) and has no error
    // channel; a missing GPU or a HIP failure is therefore reported on stderr (and through
    // lr_last_error()) before the same NULL is returned.  There is no CPU fallback.
    auto fail = []() -> LineSegment* {
        std::fprintf(stderr, "librectify_amd: find_line_segment_groups failed: %s\n", get_error().c_str());
        return nullptr;
    };
    try {
        if (buffer == nullptr || width <= 0 || height <= 0) {
            set_error("find_line_segment_groups: no image");
            return fail();
        }
        // A frame smaller than the 5x5 kernel: the reference's convolution loops do not run, there are no peaks, no lines, and
        // it returns its silent NULL (interface.cpp:50-54) -- so does this, without a word on stderr.
        if (width < 5 || height < 5) return nullptr;
        lr_context* c = thread_context();
        if (!c) return fail();
        std::vector<LineSegment> res;
        if (ctx_find_groups_host(c, buffer, width, height, stride, min_length, refine, num_threads, res)) return fail();
        if (res.empty()) return nullptr;
        LineSegment* out = new (std::nothrow) LineSegment[res.size()];
        if (!out) {
            set_error("find_line_segment_groups: out of host memory");
            return fail();
        }
        std::memcpy(out, res.data(), res.size() * sizeof(LineSegment));
        if (n_lines) *n_lines = (int)res.size();
        return out;
    } catch (...) {
        (void)guard_fail("find_line_segment_groups");
        return fail();
    }
}

void release_line_segments(LineSegment** lines) {
    if (lines && *lines != nullptr) {
        delete[] * lines;
        *lines = nullptr;
    }
}

ImageTransform compute_rectification_transform(LineSegment* lines, int n_lines, int width, int height,
                                               const RectificationConfig& cfg) {
    try {
        return rectification_transform(lines, n_lines, width, height, cfg);
    } catch (...) {  // (out of host memory: the identity corners, as for "no vanishing point")
        (void)guard_fail("compute_rectification_transform");
        return rectification_transform(nullptr, 0, width, height, cfg);
    }
}

ImageTransform compute_rectification_transform_from_vp(int width, int height, const Point& vp_h, const Point& vp_v) {
    return rectification_transform_from_vp(width, height, vp_h, vp_v);
}

Point fit_vanishing_point(const LineSegment* lines, int n_lines, int group) {
    try {
        const Vec3 v = fit_single_vanishing_point(std::vector<LineSegment>(lines, lines + n_lines), group);
        return Point{v.x, v.y, v.z};
    } catch (...) {
        (void)guard_fail("fit_vanishing_point");
        return Point{0.f, 0.f, 0.f};
    }
}

void assign_to_group(const LineSegment* lines_array, int n_lines, LineSegment* new_lines_array, int n_new_lines,
                     float angular_tolarance) {
    try {
        assign_groups(lines_array, n_lines, new_lines_array, n_new_lines, angular_tolarance);
    } catch (...) {
        (void)guard_fail("assign_to_group");
    }
}

// ---- extensions -----------------------------------------------------------------------------

int lr_context_create(int device, lr_context** out) {
    try {
        return ctx_create(device, out);
    } catch (...) {
        return guard_fail("lr_context_create");
    }
}
void lr_release_thread_context(void) { thread_context_slot().reset(); }
int lr_context_trim(lr_context* ctx) {
    try {
        return ctx_trim(ctx, true);
    } catch (...) {
        return guard_fail("lr_context_trim");
    }
}
int lr_trim_thread_context(void) {
    try {
        lr_context* c = thread_context_slot().get();
        return c ? ctx_trim(c, true) : 0;
    } catch (...) {
        return guard_fail("lr_trim_thread_context");
    }
}
void lr_context_destroy(lr_context* ctx) { ctx_destroy(ctx); }
const char* lr_last_error(void) { return get_error().c_str(); }
int lr_synchronize(lr_context* ctx) {
    try {
        LR_HIP(hipStreamSynchronize(ctx->stream));
        return 0;
    } catch (...) {
        return guard_fail("lr_synchronize");
    }
}
void lr_set_ransac_seed(lr_context* ctx, uint64_t seed) { ctx->ransac_seed = seed; }
void lr_set_stage_timing(lr_context* ctx, int on) { ctx->timing_on = on != 0; }
void lr_set_ransac_iterations(lr_context* ctx, int n_iter) { ctx->ransac_iters = n_iter; }
void lr_set_flood_mode(lr_context* ctx, int mode) {
    ctx->flood_mode = mode;
    ctx->flood_big_hint = true;  // forget what the previous frame needed
    ctx->flood_hold_hint = false;
    ctx->flood_staged_hint = false;
}
int lr_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int lr_find_line_segment_groups_device(lr_context* ctx, const float* d_image, int width, int height, int stride,
                                       float min_length, int refine, int num_threads, LineSegment* out, int capacity,
                                       int* n_lines) {
    try {
        (void)num_threads;
        std::vector<LineSegment> res;
        if (ctx_find_groups_device(ctx, d_image, width, height, stride, min_length, refine != 0, res)) return 1;
        return copy_out(res, out, capacity, n_lines);
    } catch (...) {
        return guard_fail("lr_find_line_segment_groups_device");
    }
}

int lr_find_line_segment_groups_host(lr_context* ctx, const float* buffer, int width, int height, int stride,
                                     float min_length, int refine, int num_threads, LineSegment* out, int capacity,
                                     int* n_lines) {
    try {
        if (width < 5 || height < 5 || buffer == nullptr) {
            set_error("image smaller than the 5x5 filter");
            return 1;
        }
        std::vector<LineSegment> res;
        if (ctx_find_groups_host(ctx, buffer, width, height, stride, min_length, refine != 0, num_threads, res)) return 1;
        return copy_out(res, out, capacity, n_lines);
    } catch (...) {
        return guard_fail("lr_find_line_segment_groups_host");
    }
}

int lr_find_line_segment_groups_batch_device(lr_context* ctx, const float* d_images, size_t image_stride, int batch,
                                             int width, int height, int stride, float min_length, int refine,
                                             int num_threads, LineSegment* out, int capacity, int* n_lines,
                                             const RectificationConfig* cfg, ImageTransform* transforms) {
    try {
        (void)num_threads;
        return ctx_find_groups_batch_device(ctx, d_images, image_stride, batch, width, height, stride, min_length,
                                            refine != 0, out, capacity, n_lines, cfg, transforms);
    } catch (...) {
        return guard_fail("lr_find_line_segment_groups_batch_device");
    }
}

int lr_find_line_segment_groups_batch_host(lr_context* ctx, const float* frames, size_t image_stride, int batch,
                                           int width, int height, int stride, float min_length, int refine,
                                           int num_threads, LineSegment* out, int capacity, int* n_lines,
                                           const RectificationConfig* cfg, ImageTransform* transforms) {
    try {
        std::vector<const float*> ptrs((size_t)std::max(batch, 0));
        for (int b = 0; b < batch; ++b) ptrs[b] = frames + (size_t)b * image_stride;
        return ctx_find_groups_batch_host(ctx, ptrs.data(), batch, width, height, stride, min_length, refine != 0,
                                          num_threads, out, capacity, n_lines, cfg, transforms);
    } catch (...) {
        return guard_fail("lr_find_line_segment_groups_batch_host");
    }
}

int lr_find_line_segment_groups_batch_host_ptrs(lr_context* ctx, const float* const* frames, int batch, int width,
                                                int height, int stride, float min_length, int refine, int num_threads,
                                                LineSegment* out, int capacity, int* n_lines,
                                                const RectificationConfig* cfg, ImageTransform* transforms) {
    try {
        return ctx_find_groups_batch_host(ctx, frames, batch, width, height, stride, min_length, refine != 0, num_threads,
                                          out, capacity, n_lines, cfg, transforms);
    } catch (...) {
        return guard_fail("lr_find_line_segment_groups_batch_host_ptrs");
    }
}

int lr_find_line_segment_groups_batch_host_multi(lr_context* ctx, const int* devices, int n_devices,
                                                 const float* const* frames, int batch, int width, int height, int stride,
                                                 float min_length, int refine, int num_threads, LineSegment* out,
                                                 int capacity, int* n_lines, const RectificationConfig* cfg,
                                                 ImageTransform* transforms) {
    try {
        return ctx_find_groups_batch_host_multi(ctx, devices, n_devices, frames, batch, width, height, stride, min_length,
                                                refine != 0, num_threads, out, capacity, n_lines, cfg, transforms);
    } catch (...) {
        return guard_fail("lr_find_line_segment_groups_batch_host_multi");
    }
}

int lr_host_alloc(lr_context* ctx, size_t bytes, void** out) {
    try {
        LR_HIP(hipSetDevice(ctx->device));
        LR_HIP(hipHostMalloc(out, bytes));
        return 0;
    } catch (...) {
        return guard_fail("lr_host_alloc");
    }
}
int lr_host_free(lr_context* ctx, void* p) {
    try {
        LR_HIP(hipSetDevice(ctx->device));
        LR_HIP(hipHostFree(p));
        return 0;
    } catch (...) {
        return guard_fail("lr_host_free");
    }
}

// minimal device-memory helpers so that C callers (and the tests) need no other runtime binding
int lr_device_malloc(lr_context* ctx, size_t bytes, void** out) {
    try {
        LR_HIP(hipSetDevice(ctx->device));
        LR_HIP(hipMalloc(out, bytes));
        return 0;
    } catch (...) {
        return guard_fail("lr_device_malloc");
    }
}
int lr_device_free(lr_context* ctx, void* p) {
    try {
        LR_HIP(hipSetDevice(ctx->device));
        LR_HIP(hipFree(p));
        return 0;
    } catch (...) {
        return guard_fail("lr_device_free");
    }
}
int lr_memcpy_h2d(lr_context* ctx, void* dst, const void* src, size_t bytes) {
    try {
        LR_HIP(hipSetDevice(ctx->device));
        LR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
        return 0;
    } catch (...) {
        return guard_fail("lr_memcpy_h2d");
    }
}

void lr_set_batch_streams(lr_context* ctx, int n) { ctx->batch_streams = n < 1 ? 1 : n; }
void lr_set_seed_capacity(lr_context* ctx, uint32_t cap) { ctx->seed_cap_once = cap; }
void lr_set_flood_staged(lr_context* ctx, int on) { ctx->flood_staged = on != 0; }
void lr_set_flood_blind_rounds(lr_context* ctx, int rounds) { ctx->flood_rounds_hint = rounds; }
void lr_set_flood_partial_commits(lr_context* ctx, int on) { ctx->flood_partial = on != 0; }
void lr_set_flood_multi_source(lr_context* ctx, int on) { ctx->flood_multi = on != 0; }
void lr_set_flood_just_in_time(lr_context* ctx, int on) { ctx->flood_jit = on != 0; }
void lr_set_flood_giant_step(lr_context* ctx, int on) { ctx->flood_giant_step = on != 0; }
void lr_set_flood_logs(lr_context* ctx, int on) {
    ctx->flood_logs = on != 0;
    ctx->flood_log_sweep = on == 2;
}

int lr_stage_filter(lr_context* ctx, const float* d_image, int width, int height, int stride) {
    try {
        return ctx_stage_filter(ctx, d_image, width, height, stride);
    } catch (...) {
        return guard_fail("lr_stage_filter");
    }
}
int lr_stage_filter_host(lr_context* ctx, const float* buffer, int width, int height, int stride) {
    try {
        if (width < 5 || height < 5 || buffer == nullptr) {
            set_error("image smaller than the 5x5 filter");
            return 1;
        }
        if (upload_host_image(ctx, buffer, width, height, stride, -1)) return 1;
        return ctx_stage_filter(ctx, ctx->d_img_slot[0], width, height, width);
    } catch (...) {
        return guard_fail("lr_stage_filter_host");
    }
}
int lr_stage_seeds(lr_context* ctx, int* n_seeds) {
    try {
        if (ctx_stage_seeds(ctx)) return 1;
        if (n_seeds) *n_seeds = (int)ctx->n_seeds;
        return 0;
    } catch (...) {
        return guard_fail("lr_stage_seeds");
    }
}
int lr_stage_flood(lr_context* ctx, int* n_components) {
    try {
        if (ctx_stage_flood(ctx)) return 1;
        LR_HIP(hipStreamSynchronize(ctx->stream));
        if (n_components) *n_components = -1;  // known after lr_stage_fit
        return 0;
    } catch (...) {
        return guard_fail("lr_stage_flood");
    }
}
int lr_stage_fit(lr_context* ctx, LineSegment* out, int capacity, int* n_lines) {
    try {
        std::vector<LineSegment> res;
        if (ctx_stage_fit(ctx, res)) return 1;
        return copy_out(res, out, capacity, n_lines);
    } catch (...) {
        return guard_fail("lr_stage_fit");
    }
}

int lr_download(lr_context* ctx, int buffer_id, void* dst, size_t bytes) {
    try {
        const size_t npix = (size_t)ctx->w * ctx->h;
        const void* src = nullptr;
        size_t have = 0;
        switch (buffer_id) {
            case LR_BUF_DX: src = ctx->dx; have = npix * 4; break;
            case LR_BUF_DY: src = ctx->dy; have = npix * 4; break;
            case LR_BUF_DMASK: src = ctx->dmask; have = npix; break;
            case LR_BUF_LABEL: src = ctx->label; have = npix * 4; break;
            case LR_BUF_SEED_IDX: src = ctx->seed_idx; have = (size_t)ctx->n_seeds * 4; break;
            case LR_BUF_SEED_BIN: src = ctx->seed_bin; have = (size_t)ctx->n_seeds * 4; break;
            case LR_BUF_SEED_THR: src = ctx->seed_thr; have = (size_t)ctx->n_seeds * 4; break;
            case LR_BUF_MAXMAG: src = ctx->maxmag; have = 4; break;
            case LR_BUF_SEED_SIZE: src = ctx->seed_size; have = (size_t)ctx->n_seeds * 4; break;
            default: set_error("lr_download: unknown buffer id"); return 1;
        }
        if (buffer_id == LR_BUF_DMASK && ctx->dmask_consumed) {
            set_error("lr_download: LR_BUF_DMASK was consumed by lr_stage_flood (labelled pixels are cleared); download it after lr_stage_filter");
            return 1;
        }
        if (bytes > have) {
            set_error("lr_download: request larger than the buffer");
            return 1;
        }
        LR_HIP(hipSetDevice(ctx->device));
        LR_HIP(hipStreamSynchronize(ctx->stream));
        LR_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
        return 0;
    } catch (...) {
        return guard_fail("lr_download");
    }
}

int lr_filter_kernel_ms(lr_context* ctx, float* ms) {
    try {
        LR_HIP(hipStreamSynchronize(ctx->stream));
        LR_HIP(hipEventElapsedTime(ms, ctx->ev[0], ctx->ev[1]));
        return 0;
    } catch (...) {
        return guard_fail("lr_filter_kernel_ms");
    }
}

int lr_stage_times(lr_context* ctx, float* ms, int count) {
    try {
        for (int i = 0; i < count && i < LR_T_COUNT; ++i) ms[i] = ctx->stage_ms[i];
        return 0;
    } catch (...) {
        return guard_fail("lr_stage_times");
    }
}

int lr_stage_counters(lr_context* ctx, int64_t* out, int count) {
    try {
        const int64_t v[16] = {(int64_t)ctx->n_seeds,        (int64_t)ctx->n_comp,         (int64_t)ctx->flood_rounds,
                               (int64_t)ctx->n_px,           (int64_t)ctx->flood_tiers[0], (int64_t)ctx->flood_tiers[1],
                               (int64_t)ctx->flood_tiers[2], (int64_t)ctx->frame_laps,
                               (int64_t)(((uint64_t)ctx->flood_tiers[5] << 32) | ctx->flood_tiers[4]),
                               (int64_t)(((uint64_t)ctx->flood_tiers[7] << 32) | ctx->flood_tiers[6]),
                               (int64_t)ctx->flood_tiers[9], (int64_t)ctx->flood_tiers[10], (int64_t)ctx->flood_tiers[11], (int64_t)ctx->flood_tiers[12],
                               (int64_t)ctx->flood_tiers[13], (int64_t)ctx->flood_tiers[15]};
        for (int i = 0; i < count && i < 16; ++i) out[i] = v[i];
        return 0;
    } catch (...) {
        return guard_fail("lr_stage_counters");
    }
}

int lr_ransac_best(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float tol,
                   int n_iter, uint64_t seed, uint32_t round, float* best_h3, float* best_score, int* best_iter) {
    try {
        const PencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
        const std::vector<int> idx(indices, indices + n_idx);
        Vec3 h;
        float s;
        int it;
        if (ctx_ransac_best(ctx, model, idx, tol, n_iter, seed, round, &h, &s, &it)) return 1;
        best_h3[0] = h.x;
        best_h3[1] = h.y;
        best_h3[2] = h.z;
        *best_score = s;
        *best_iter = it;
        return 0;
    } catch (...) {
        return guard_fail("lr_ransac_best");
    }
}

int lr_estimate_line_pencils(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                             float garbage_deg, int n_iter, uint64_t seed) {
    try {
        std::vector<LineSegment> v(lines, lines + n);
        if (ctx_estimate_line_pencils(ctx, v, max_models, inlier_deg, garbage_deg, n_iter, seed)) return 1;
        std::memcpy(lines, v.data(), sizeof(LineSegment) * (size_t)n);
        return 0;
    } catch (...) {
        return guard_fail("lr_estimate_line_pencils");
    }
}

int lr_cht_vanishing_point(lr_context* ctx, const LineSegment* lines, int n, int d, Point* vp, uint64_t* acc_out) {
    try {
        Vec3 p;
        std::vector<uint64_t> acc;
        if (ctx_cht_vanishing_point(ctx, std::vector<LineSegment>(lines, lines + n), d, &p, acc_out ? &acc : nullptr)) return 1;
        *vp = Point{p.x, p.y, p.z};
        if (acc_out) std::memcpy(acc_out, acc.data(), acc.size() * sizeof(uint64_t));
        return 0;
    } catch (...) {
        return guard_fail("lr_cht_vanishing_point");
    }
}

int lr_refine_lines(lr_context* ctx, const LineSegment* in, int n, LineSegment* out, int* n_out) {
    try {
        std::vector<LineSegment> v(in, in + n);
        if (ctx_refine(ctx, v)) return 1;
        std::memcpy(out, v.data(), v.size() * sizeof(LineSegment));
        *n_out = (int)v.size();
        return 0;
    } catch (...) {
        return guard_fail("lr_refine_lines");
    }
}

void lr_set_estimator(lr_context* ctx, int kind, int param) {
    ctx->estimator = kind;
    if (kind == 3) ctx->cht_d = param > 0 ? param : 128;
    else ctx->prosac_T_N = param;
}

int lr_estimate_line_pencils_cht(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                                 float garbage_deg, int d, float* models3, int* n_models, uint32_t* peak_cells,
                                 uint64_t* votes) {
    try {
        std::vector<LineSegment> v(lines, lines + n);
        ChtTrace tr;
        if (ctx_estimate_line_pencils_cht(ctx, v, max_models, inlier_deg, garbage_deg, d, &tr)) return 1;
        std::memcpy(lines, v.data(), sizeof(LineSegment) * (size_t)n);
        for (size_t k = 0; k < tr.models.size(); ++k) {
            if (models3) {
                models3[3 * k + 0] = tr.models[k].x;
                models3[3 * k + 1] = tr.models[k].y;
                models3[3 * k + 2] = tr.models[k].z;
            }
            if (peak_cells) peak_cells[k] = tr.peak_cell[k];
        }
        if (n_models) *n_models = (int)tr.models.size();
        if (votes) *votes = tr.votes;
        return 0;
    } catch (...) {
        return guard_fail("lr_estimate_line_pencils_cht");
    }
}

int lr_ht_weights(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* weights) {
    try {
        const PencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
        std::vector<float> w;
        if (ctx_ht_weights(ctx, model, std::vector<int>(indices, indices + n_idx), w)) return 1;
        std::memcpy(weights, w.data(), w.size() * sizeof(float));
        return 0;
    } catch (...) {
        return guard_fail("lr_ht_weights");
    }
}

int lr_prosac_solve(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float tol,
                    int T_N, uint64_t seed, uint32_t round, float* h3, int32_t* trace4) {
    try {
        const PencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
        Vec3 h;
        ProsacTrace tr;
        if (ctx_prosac_solve(ctx, model, std::vector<int>(indices, indices + n_idx), tol, T_N, seed, round, &h, &tr)) return 1;
        h3[0] = h.x;
        h3[1] = h.y;
        h3[2] = h.z;
        if (trace4) {
            trace4[0] = tr.iterations;
            trace4[1] = tr.n_star;
            trace4[2] = tr.best_iter;
            trace4[3] = tr.I_N_best;
        }
        return 0;
    } catch (...) {
        return guard_fail("lr_prosac_solve");
    }
}

int lr_direct_solve(lr_context* ctx, const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* h3) {
    try {
        const PencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
        Vec3 h;
        if (ctx_direct_solve(ctx, model, std::vector<int>(indices, indices + n_idx), &h)) return 1;
        h3[0] = h.x;
        h3[1] = h.y;
        h3[2] = h.z;
        return 0;
    } catch (...) {
        return guard_fail("lr_direct_solve");
    }
}

int lr_estimate_line_pencils_direct(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                                    float garbage_deg) {
    try {
        std::vector<LineSegment> v(lines, lines + n);
        if (ctx_estimate_line_pencils_direct(ctx, v, max_models, inlier_deg, garbage_deg)) return 1;
        std::memcpy(lines, v.data(), sizeof(LineSegment) * (size_t)n);
        return 0;
    } catch (...) {
        return guard_fail("lr_estimate_line_pencils_direct");
    }
}

int lr_estimate_line_pencils_prosac(lr_context* ctx, LineSegment* lines, int n, int max_models, float inlier_deg,
                                    float garbage_deg, int T_N, uint64_t seed) {
    try {
        std::vector<LineSegment> v(lines, lines + n);
        if (ctx_estimate_line_pencils_prosac(ctx, v, max_models, inlier_deg, garbage_deg, T_N, seed)) return 1;
        std::memcpy(lines, v.data(), sizeof(LineSegment) * (size_t)n);
        return 0;
    } catch (...) {
        return guard_fail("lr_estimate_line_pencils_prosac");
    }
}

}  // extern "C"
