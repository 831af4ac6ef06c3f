// Host pipeline: owns the workspace, enqueues the stages on the context stream, and runs the
// (sequential, tiny) vanishing-point peeling loop around the GPU scoring kernel.
#include "context.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <map>
#include <mutex>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <pthread.h>
#include <sched.h>
#if defined(__SSE2__)
#include <immintrin.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif


// Copy into a page-locked staging buffer with non-temporal stores (SSE2, part of every x86-64): a plain memcpy reads the
// destination's lines before it overwrites them, and the staging buffers are read next by the DMA engine, not by a core.
// Less host-memory traffic beside the transfers, which read the same memory.
static inline void stage_copy(void* dst, const void* src, size_t n) {
#if defined(__SSE2__)
    static const bool plain = std::getenv("LIBRECTIFY_STAGE_PLAIN") != nullptr;  // (comparison knob)
    char* d = static_cast<char*>(dst);
    const char* s_ = static_cast<const char*>(src);
    if (plain || n < 4096) {
        std::memcpy(d, s_, n);
        return;
    }
    const size_t head = (16u - (reinterpret_cast<uintptr_t>(d) & 15u)) & 15u;
    if (head) {
        std::memcpy(d, s_, head);
        d += head;
        s_ += head;
        n -= head;
    }
    size_t blocks = n / 64;
    while (blocks--) {
        const __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s_));
        const __m128i b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s_ + 16));
        const __m128i c2 = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s_ + 32));
        const __m128i e = _mm_loadu_si128(reinterpret_cast<const __m128i*>(s_ + 48));
        _mm_stream_si128(reinterpret_cast<__m128i*>(d), a);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 16), b);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 32), c2);
        _mm_stream_si128(reinterpret_cast<__m128i*>(d + 48), e);
        s_ += 64;
        d += 64;
    }
    n &= 63;
    if (n) std::memcpy(d, s_, n);
    _mm_sfence();
#else
    std::memcpy(dst, src, n);
#endif
}

namespace lramd {

namespace {

thread_local std::string g_error;

template <class T>
int dev_alloc(T*& p, size_t count) {
    if (p) {
        (void)hipFree(p);
        p = nullptr;
    }
    LR_HIP(hipMalloc((void**)&p, std::max<size_t>(count, 1) * sizeof(T)));
    return 0;
}

// 1-D factors of the reference's taps (filter.cpp:65-78: H = z / a * exp(-(x^2 + y^2) / 2s^2), a = 2 pi s^4):
// d(t) = t / a * exp(-t^2 / 2s^2), g(t) = exp(-t^2 / 2s^2), evaluated with the same float operations
void gauss_deriv_factors(int size, float sigma, float* d, float* g) {
    const int n = 2 * size + 1;
    for (int t = 0; t < n; ++t) {
        const float x = float(t - size);
        const float a = float(2 * M_PI * std::pow(sigma, 4.0f));
        const float e = std::exp(-std::pow(x, 2.0f) / (2 * std::pow(sigma, 2.0f)));
        d[t] = x / a * e;
        g[t] = e;
    }
}

void init_constants(lr_context* c) {
    gauss_deriv_factors(kEdgeKernelSize, kEdgeKernelSigma, c->fconsts.d, c->fconsts.g);
    for (int b = 0; b < kBins; ++b) {
        const float theta = float(b * M_PI) / kBins;  // line_detector.cpp:144
        c->trig.st[b] = std::sin(theta);
        c->trig.ct[b] = std::cos(theta);
        c->fconsts.st[b] = c->trig.st[b];
        c->fconsts.ct[b] = c->trig.ct[b];
    }
    c->seed_keep_ratio = 1 - std::max(std::min(kSeedRatio, 1.f), 0.f);  // line_detector.cpp:209
}

int idx_bits_for(size_t npix) {
    int b = 1;
    while (((size_t)1 << b) < npix) ++b;
    return b;
}

}  // namespace

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void set_error(const std::string& msg) { g_error = msg; }
const std::string& get_error() { return g_error; }

// Gives the device and page-locked memory that is sized by the largest frame the context has seen back to the system: the
// per-pixel workspace (about 130 bytes a pixel), the flood's per-seed buffers, slabs and hand-over records, the frame slots
// and their staging buffers, the batch ring.  Streams, events and small tables stay; the next frame allocates what it needs.
int ctx_trim(lr_context* c, bool frames_too) {
    LR_HIP(hipSetDevice(c->device));
    LR_HIP(hipStreamSynchronize(c->stream));
    if (c->copy_stream) LR_HIP(hipStreamSynchronize(c->copy_stream));
    if (c->flood_aux) LR_HIP(hipStreamSynchronize(c->flood_aux));
    auto drop = [](auto*& p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    };
    drop(c->dx), drop(c->dy), drop(c->dmask), drop(c->cand), drop(c->cand_count), drop(c->tile_max), drop(c->tile_pass), drop(c->tile_off);
    drop(c->keys_a), drop(c->keys_b), drop(c->seed_idx), drop(c->seed_bin), drop(c->seed_thr), drop(c->seed_size), drop(c->label), drop(c->queue);
    drop(c->comp_rank), drop(c->comp_seed), drop(c->comp_off), drop(c->cursor), drop(c->px_a), drop(c->px_b), drop(c->scratch_w), drop(c->d_lines);
    drop(c->comp_large), drop(c->huge.tab), drop(c->huge.jobs), drop(c->huge.list), drop(c->temp);
    c->temp_bytes = 0;
    c->cap_pix = 0;
    c->cap_tiles = 0;
    FloodBuffers& f = c->fb;
    drop(f.blocked), drop(f.count), drop(f.flags), drop(f.state), drop(f.tier), drop(f.blk), drop(f.act_a), drop(f.act_b), drop(f.ctrl), drop(f.big_list);
    drop(f.handover), drop(f.dirty), drop(f.giant_mask), drop(f.waypoints), drop(f.multi_list), drop(f.log_off), drop(f.log_len), drop(f.log_buf);
    drop(f.slab_ring), drop(f.slab_hash);
    c->fb_cap_seeds = 0;
    // (the frame slots hold the frame that is being processed when the workspace shrinks by itself: only on request)
    for (int s = 0; s < 2 && frames_too; ++s) {
        drop(c->d_img_slot[s]);
        c->cap_slot[s] = 0;
        if (c->h_stage[s]) (void)hipHostFree(c->h_stage[s]);
        c->h_stage[s] = nullptr;
        c->cap_stage[s] = 0;
    }
    if (frames_too) {
        for (float*& p : c->ring_img) drop(p);
        c->ring_cap_pix = 0;
        for (float*& p : c->ring_stage) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
        }
        c->ring_stage_cap_pix = 0;
    }
    c->small_frames = 0;
    c->w = c->h = 0;
    c->seed_cap = 0;
    c->flood_hold_hint = c->flood_staged_hint = c->flood_calm_hint = false;  // (what the frames before taught the context goes with the workspace)
    c->flood_staged_streak = 0;
    for (bool& v : c->stage_valid) v = false;
    // (the lanes of a batch call are contexts of their own that shrink by themselves; a lane that shrinks in the middle of a
    // call must not touch the others, which are in the middle of their frames -- only a trim on request goes through them)
    if (frames_too)
        for (lr_context* wc : c->workers)
            if (ctx_trim(wc, true)) return 1;
    return 0;
}

int ctx_ensure_image_capacity(lr_context* c, int w, int h) {
    const size_t npix = (size_t)w * h;
    // What a context hands from frame to frame describes the previous frame of the SAME stream: a frame of another size is the
    // start of another one (bench.py's content-latency frames: the 1080p soft blobs inherited "staged" from the 4K ones and
    // kept it -- 17 ms a frame instead of 13.7).
    if (c->w != 0 && (c->w != w || c->h != h)) {
        c->flood_hold_hint = false;
        c->flood_staged_hint = false;
        c->flood_calm_hint = false;
        c->flood_staged_streak = 0;
    }
    const FilterGeom fg = filter_geometry(w, h);
    const int ntiles = fg.n_tiles;
    // A context that has seen one large frame keeps serving small ones out of the large workspace (an 8192 x 8192 call leaves
    // 8.9 GB behind): after eight frames in a row of at most a quarter of the capacity it is given back, and this frame
    // allocates its own size (lr_context_trim does the same at once).
    if (c->cap_pix != 0 && npix * 4 <= c->cap_pix) {
        if (++c->small_frames >= 8 && ctx_trim(c, false)) return 1;
    } else {
        c->small_frames = 0;
    }
    if (npix <= c->cap_pix && ntiles <= c->cap_tiles) return 0;
    LR_HIP(hipStreamSynchronize(c->stream));
    const size_t cp = std::max(npix, c->cap_pix);
    const int ct = std::max(ntiles, c->cap_tiles);
    if (dev_alloc(c->dx, cp) || dev_alloc(c->dy, cp) || dev_alloc(c->dmask, cp + 16) ||
        dev_alloc(c->cand, (size_t)ct * fg.cand_cap) || dev_alloc(c->cand_count, ct) || dev_alloc(c->tile_max, ct) ||
        dev_alloc(c->tile_pass, ct) || dev_alloc(c->tile_off, ct) || dev_alloc(c->keys_a, cp) ||
        dev_alloc(c->keys_b, cp) || dev_alloc(c->seed_idx, cp) || dev_alloc(c->seed_bin, cp) ||
        dev_alloc(c->seed_thr, cp) || dev_alloc(c->seed_size, cp) || dev_alloc(c->label, cp) ||
        dev_alloc(c->queue, cp) || dev_alloc(c->comp_rank, cp) || dev_alloc(c->comp_seed, cp) ||
        dev_alloc(c->comp_off, cp + 1) || dev_alloc(c->cursor, cp) || dev_alloc(c->px_a, cp) ||
        dev_alloc(c->px_b, cp) || dev_alloc(c->scratch_w, cp) || dev_alloc(c->d_lines, cp / 6 + 16) ||
        dev_alloc(c->comp_large, cp / 64 + 16))
        return 1;
    LR_HIP(hipMemsetAsync(c->tile_off, 0, (size_t)ct * sizeof(uint32_t), c->stream));  // (seed_select_kernel's status words)
    {   // huge components: a row of buckets (2^14 pixel indices each) for up to 256 of them; the table is zero between frames
        const size_t nb = (cp + 16383) >> 14;
        c->huge.max = 256;
        if (dev_alloc(c->huge.tab, c->huge.max * nb) || dev_alloc(c->huge.jobs, 4 * c->huge.max * nb) || dev_alloc(c->huge.list, c->huge.max)) return 1;
        LR_HIP(hipMemsetAsync(c->huge.tab, 0, c->huge.max * nb * sizeof(uint32_t), c->stream));
    }
    const size_t tb = std::max(seeds_temp_bytes(ct, cp), fit_temp_bytes(cp, (uint32_t)std::min<size_t>(cp / 6 + 16, 0xFFFFFFFFu)));
    if (c->temp) (void)hipFree(c->temp);
    c->temp = nullptr;
    LR_HIP(hipMalloc(&c->temp, tb));
    LR_HIP(hipMemsetAsync(c->temp, 0, tb, c->stream));  // (component_offsets_kernel's status words)
    c->temp_bytes = tb;
    c->cap_pix = cp;
    c->cap_tiles = ct;
    return 0;
}

// The overflow slabs (2.25 MB each).  With the giant step the default mode needs them for nothing but the walks of the
// lowest active seed when the step is switched off: 16 of them (36 MB; the default pool was 128 = 288 MB).  The storage
// test hooks (modes 2-7: no second tier, every long walk in a slab) and LIBRECTIFY_FLOOD_GIANT_STEP=0 get the full pool.
static int ensure_flood_slabs(lr_context* c) {
    FloodBuffers& f = c->fb;
    static const bool giant_step_off = std::getenv("LIBRECTIFY_FLOOD_GIANT_STEP") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_GIANT_STEP")) == 0;
    uint32_t want = (c->flood_mode == 1 && c->flood_giant_step && !giant_step_off) ? 16u : 128u;
    if (const char* e = std::getenv("LIBRECTIFY_FLOOD_SLABS")) want = (uint32_t)std::max(1, std::atoi(e));
    if (f.slab_ring && f.n_slabs >= want) return 0;
    LR_HIP(hipStreamSynchronize(c->stream));
    f.n_slabs = want;
    f.slab_ring_cap = 1u << 14;  // (tile, entry mask) records, 16 B each
    f.slab_hash_cap = 1u << 16;  // tile -> (walked, acceptable) records, 32 B each (48 Ki tiles = 3 M pixels)
    uint4* r = (uint4*)f.slab_ring;
    uint4* hsh = (uint4*)f.slab_hash;
    if (dev_alloc(r, (size_t)f.n_slabs * f.slab_ring_cap) || dev_alloc(hsh, (size_t)f.n_slabs * f.slab_hash_cap * 2)) return 1;
    f.slab_ring = r;
    f.slab_hash = hsh;
    LR_HIP(hipMemsetAsync(f.slab_hash, 0, (size_t)f.n_slabs * f.slab_hash_cap * 32, c->stream));
    if (f.ctrl) LR_HIP(hipMemsetAsync(f.ctrl, 0, kFloodCtrlWords * sizeof(uint32_t), c->stream));  // (the generation counter starts again with the cleared tables)
    return 0;
}

static int ensure_flood_buffers(lr_context* c) {
    FloodBuffers& f = c->fb;
    if (c->fb_cap_seeds >= c->cap_pix && f.ctrl) return ensure_flood_slabs(c);
    LR_HIP(hipStreamSynchronize(c->stream));
    const size_t cs = c->cap_pix;
    if (dev_alloc(f.blocked, cs) || dev_alloc(f.count, cs) || dev_alloc(f.flags, cs) || dev_alloc(f.state, cs) || dev_alloc(f.tier, cs) || dev_alloc(f.blk, cs) ||
        dev_alloc(f.act_a, cs) || dev_alloc(f.act_b, cs) || dev_alloc(f.ctrl, kFloodCtrlWords) || dev_alloc(f.big_list, 8192) || dev_alloc(f.handover, 8192 * kFloodHandWords) ||
        dev_alloc(f.dirty, cs / 256 + 16))
        return 1;
    {   // the giant step's tile masks: (w + 7) / 8 x (h + 7) / 8 tiles <= cs / 16 + 64 for frames of 5 x 5 and more
        uint64_t* gm = (uint64_t*)f.giant_mask;
        if (dev_alloc(gm, cs / 16 + 64)) return 1;
        f.giant_mask = gm;
    }
    f.wp_cap = (uint32_t)std::min<size_t>(std::max<size_t>(cs / 16, 4096), cs);  // (one seed per 16 pixels: the 4K bench frame has one per 200)
    if (dev_alloc(f.waypoints, (size_t)f.wp_cap * kFloodWpWords) || dev_alloc(f.multi_list, 8192)) return 1;
    // footprint logs (FloodBuffers::rewalk_logs): per-seed words for one seed per 16 pixels, a record per 8 pixels
    f.log_seeds = f.wp_cap;
    static const int log_div = std::getenv("LIBRECTIFY_FLOOD_LOG_CAP_DIV") ? std::max(1, std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOG_CAP_DIV"))) : 8;
    f.log_cap = (uint32_t)std::min<size_t>(std::max<size_t>(cs / (size_t)log_div, 65536), 1u << 28);
    if (dev_alloc(f.log_off, f.log_seeds) || dev_alloc(f.log_len, f.log_seeds) || dev_alloc(f.log_buf, (size_t)f.log_cap * 3)) return 1;
    if (!c->flood_aux) {
        // At a priority of its own: HIP maps streams onto a few hardware queues, and a second stream that lands on the queue
        // of the first runs BEHIND it -- fork and join then cost two barriers a round and buy nothing (seen in bench.py, whose
        // process has thirteen streams: flood 1.28 -> 1.80 ms).  Streams of different priorities never share a queue.
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
        if (hi == lo) LR_HIP(hipStreamCreateWithFlags(&c->flood_aux, hipStreamNonBlocking));
        else LR_HIP(hipStreamCreateWithPriority(&c->flood_aux, hipStreamNonBlocking, hi));
        for (int i = 0; i < 8; ++i) {
            hipEvent_t a = nullptr, b = nullptr;
            LR_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
            c->flood_fork.push_back(a);
            LR_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
            c->flood_join.push_back(b);
        }
    }
    LR_HIP(hipMemsetAsync(f.ctrl, 0, kFloodCtrlWords * sizeof(uint32_t), c->stream));
    c->fb_cap_seeds = cs;
    return ensure_flood_slabs(c);
}

int ctx_ensure_ransac_capacity(lr_context* c, size_t n_lines, size_t n_iter) {
    if (n_lines > c->cap_lines) {
        LR_HIP(hipStreamSynchronize(c->stream));
        const size_t cl = std::max<size_t>(n_lines, 4096);
        if (dev_alloc(c->d_model, 8 * cl)) return 1;
        if (c->h_model) (void)hipHostFree(c->h_model);
        c->h_model = nullptr;
        LR_HIP(hipHostMalloc((void**)&c->h_model, 8 * cl * sizeof(float)));
        c->cap_lines = cl;
    }
    (void)n_iter;  // (scores are not stored any more: kernels_ransac.hip)
    return 0;
}

// ---- host frames -> device slots -------------------------------------------------------------

namespace {

bool is_page_locked(const void* p) {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) {
        (void)hipGetLastError();  // an ordinary malloc'ed pointer: not an error of ours
        return false;
    }
    return a.type == hipMemoryTypeHost;
}

int staging_threads(int num_threads, size_t frame_bytes = 0) {
    // reference threading.h:24-27: t < 0 is the serial mode, otherwise min(t, available) threads (t = 0 is
    // ill-defined there: one thread here).  Eight threads saturate the copy into the staging buffer of a 4K frame; a frame
    // of more than 64 MB (8192^2: 268 MB) is one long copy in front of one long transfer, and sixteen get its first bands
    // onto the link sooner (round 4).
    if (num_threads <= 1) return 1;
    const int hw = (int)std::max(1u, std::thread::hardware_concurrency());
    return std::min(std::min(num_threads, hw), frame_bytes > ((size_t)64 << 20) ? 16 : 8);
}

// The host cores next to a device: those of the NUMA node its PCI function sits on (sysfs), as far as this process may use
// them; empty if unknown.  Staging helpers bind themselves there: copies by cores of the other socket reach 41 GB/s where
// the same copies by cores of the device's own node keep the link at 54 (tools/ubench/h2d_placement.hip,
// profiles/r04_h2d_paths.txt).  LIBRECTIFY_STAGING_BIND=0 leaves the helpers where the scheduler puts them.
// (Returned by value, copied under the lock: a reference into the cache dangled when another thread asked for a device with
// a higher index and the outer vector grew -- the start-up pattern of the multi-device batch call.)
std::vector<int> device_node_cpus(int device) {
    static std::mutex mu;
    static std::map<int, std::vector<int>> cache;
    std::lock_guard<std::mutex> lk(mu);
    const auto found = cache.find(device);
    if (found != cache.end()) return found->second;
    std::vector<int>& out = cache[device];
    static const bool off = std::getenv("LIBRECTIFY_STAGING_BIND") && std::atoi(std::getenv("LIBRECTIFY_STAGING_BIND")) == 0;
    char bus[64] = {0};
    if (off || hipDeviceGetPCIBusId(bus, (int)sizeof(bus), device) != hipSuccess) return out;
    for (char* p = bus; *p; ++p) *p = (char)std::tolower((unsigned char)*p);
    int node = -1;
    if (FILE* f = std::fopen((std::string("/sys/bus/pci/devices/") + bus + "/numa_node").c_str(), "r")) {
        if (std::fscanf(f, "%d", &node) != 1) node = -1;
        std::fclose(f);
    }
    if (node < 0) return out;
    char line[4096] = {0};
    if (FILE* f = std::fopen(("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist").c_str(), "r")) {
        if (!std::fgets(line, (int)sizeof(line), f)) line[0] = 0;
        std::fclose(f);
    }
    cpu_set_t allowed;
    CPU_ZERO(&allowed);
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return out;
    for (const char* p = line; *p;) {  // "0-63,128-191"
        char* e = nullptr;
        const long a = std::strtol(p, &e, 10);
        if (e == p) break;
        long b = a;
        if (*e == '-') b = std::strtol(e + 1, &e, 10);
        for (long cpu = a; cpu <= b && cpu < CPU_SETSIZE; ++cpu)
            if (CPU_ISSET((int)cpu, &allowed)) out.push_back((int)cpu);
        p = (*e == ',') ? e + 1 : e;
        if (*e != ',') break;
    }
    return out;
}

void bind_this_thread_near(int device) {
    const std::vector<int> cpus = device_node_cpus(device);
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    for (int c : cpus) CPU_SET(c, &set);
    (void)pthread_setaffinity_np(pthread_self(), sizeof(set), &set);  // (best effort)
}

}  // namespace

// The copy stream of a context is made when it first uploads a frame: HIP maps streams onto a few hardware queues
// (GPU_MAX_HW_QUEUES), commands of streams that share one run in order, and a batch's lanes never upload -- their
// copy streams would only take queues away from the one that does (transfers were seen waiting 4-6 ms behind another
// lane's kernels).
static int ensure_copy_stream(lr_context* c) {
    if (c->copy_stream) return 0;
    LR_HIP(hipSetDevice(c->device));
    // ... and at a higher priority than the lanes' streams: streams of different priorities do not share a queue
    int lo = 0, hi = 0;
    if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) lo = hi = 0;
    static const bool plain = std::getenv("LIBRECTIFY_COPY_STREAM_PLAIN") != nullptr;  // (measurement)
    if (plain || hi == lo) LR_HIP(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
    else LR_HIP(hipStreamCreateWithPriority(&c->copy_stream, hipStreamNonBlocking, hi));
    return 0;
}

// Rows of a host frame (|stride| >= w; a negative stride addresses the same rows from the other end, reference
// image.cpp:11-19) into a device buffer of w x h floats, on stream `up`.  Page-locked memory goes as it lies; pageable
// memory goes through the page-locked buffer `stage` in 4 MB bands, copied by up to `num_threads` threads, each band's
// transfer enqueued as soon as it is staged.  Nothing here waits for the transfers.
static int upload_rows(lr_context* c, float* dst, float* stage, const float* buffer, int w, int h, int stride, int num_threads,
                       hipStream_t up) {
    if (stride < 0) {
        buffer = buffer + (std::ptrdiff_t)(h - 1) * stride;
        stride = -stride;
    }
    const size_t npix = (size_t)w * h;
    const size_t row_bytes = (size_t)w * sizeof(float);
    if (stage == nullptr) {
        if (stride == w)  // one linear transfer: a pitched copy of the same bytes goes row by row
            LR_HIP(hipMemcpyAsync(dst, buffer, npix * sizeof(float), hipMemcpyHostToDevice, up));
        else
            LR_HIP(hipMemcpy2DAsync(dst, row_bytes, buffer, (size_t)stride * sizeof(float), row_bytes, (size_t)h,
                                    hipMemcpyHostToDevice, up));
        return 0;
    }
    const int rows_per_band = (int)std::max<size_t>(1, ((size_t)4 << 20) / row_bytes);
    const int n_bands = (h + rows_per_band - 1) / rows_per_band;
    const int T = std::min(staging_threads(num_threads), n_bands);
    std::vector<int> rc(T, 0);
    auto run = [&](int t) {
        if (t > 0 && hipSetDevice(c->device) != hipSuccess) {
            rc[t] = 1;
            return;
        }
        for (int k = t; k < n_bands; k += T) {
            const int r0 = k * rows_per_band, r1 = std::min(h, r0 + rows_per_band);
            if (stride == w) {
                std::memcpy(stage + (size_t)r0 * w, buffer + (size_t)r0 * stride, (size_t)(r1 - r0) * row_bytes);
            } else {
                for (int r = r0; r < r1; ++r) std::memcpy(stage + (size_t)r * w, buffer + (size_t)r * stride, row_bytes);
            }
            if (hipMemcpyAsync(dst + (size_t)r0 * w, stage + (size_t)r0 * w, (size_t)(r1 - r0) * row_bytes,
                               hipMemcpyHostToDevice, up) != hipSuccess)
                rc[t] = 1;
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(run, t);
    run(0);
    for (auto& x : th) x.join();
    for (int t = 0; t < T; ++t)
        if (rc[t]) {
            (void)hipGetLastError();
            set_error("upload: staging copy failed");
            return 1;
        }
    return 0;
}

// The threads that stage pageable frames for a batch call: started once per call, not once per frame (seven thread
// starts a frame were a tenth of a millisecond of the uploader's time and the larger part of its jitter).  A job is one
// frame; its 4 MB bands are claimed through a counter that carries the job's number, so that a thread that is late for
// one job cannot take a band of the next with the old job's pointers.
struct StagingCrew {
    lr_context* c = nullptr;
    std::vector<std::thread> th;
    std::atomic<uint32_t> job{0};
    std::atomic<uint64_t> next{0};   // job number << 32 | next piece
    std::atomic<uint64_t> total{0};  // job number << 32 | pieces of that job
    std::atomic<int> bands_left{0}, failed{0};
    std::atomic<bool> quit{false};
    // Two job descriptors, used alternately (job number & 1): the one a late helper may still be reading is not the one
    // the uploader fills for the next frame, and the one after that is only filled when every band of this one is done.
    struct Job {
        float* dst = nullptr;
        float* stage = nullptr;
        const float* src = nullptr;
        int w = 0, h = 0, stride = 0, rows_per_band = 1, n_bands = 0;
        hipStream_t up = nullptr;
        int pieces = 1;                       // row runs a band is copied in (by different threads)
        hipEvent_t* band_ev = nullptr;        // optional: recorded after each band's transfer is enqueued ...
        std::atomic<int>* ready = nullptr;    // ... and then ready[k] = 1 (-1 if the band failed)
    } jobs[2];

    // A single frame's bands are staged by ALL the threads together, piece by piece (`pieces` row runs per band, claimed in
    // order through `next`), and sent by whichever thread finishes a band's last piece: the first transfer starts after one
    // band's worth of copying spread over the crew instead of after every thread has copied a whole band of its own (which
    // is when all of them are ready at once).  A batch's uploader keeps whole bands per thread (pieces = 1): there the
    // link is busy with the previous frame anyway, and fewer hand-overs are worth more than an early start.
    static constexpr int kPieces = 8;
    static constexpr int kMaxBands = 256;
    std::atomic<int> pieces_left[2][kMaxBands];

    // One piece of job `gen`, if there is one left: true if a piece was claimed (and copied).  The descriptor is read only
    // AFTER the claim: a claimed piece keeps bands_left above zero, the uploader is then still inside finish() of this very
    // job, and nobody writes either descriptor (begin() of the next job comes after that finish(); the job after it, which
    // reuses this slot, after the next one's).  A helper that is late for a job finds another job's number in `next` /
    // `total` and leaves without having looked at anything else.
    bool work_one(uint32_t gen) {
        for (;;) {
            uint64_t x = next.load(std::memory_order_acquire);
            const uint64_t t = total.load(std::memory_order_acquire);
            if ((uint32_t)(x >> 32) != gen || (uint32_t)(t >> 32) != gen || (uint32_t)x >= (uint32_t)t) return false;
            if (!next.compare_exchange_weak(x, x + 1, std::memory_order_acq_rel)) continue;
            const Job& j = jobs[gen & 1u];
            const size_t row_bytes = (size_t)j.w * sizeof(float);
            const int kP = j.pieces;
            const int rows_per_piece = (j.rows_per_band + kP - 1) / kP;
            const int k = (int)(uint32_t)x / kP, piece = (int)(uint32_t)x % kP;
            const int b0 = k * j.rows_per_band, b1 = std::min(j.h, b0 + j.rows_per_band);
            const int r0 = std::min(b1, b0 + piece * rows_per_piece), r1 = std::min(b1, r0 + rows_per_piece);
            if (r1 > r0) {
                if (j.stride == j.w) {
                    stage_copy(j.stage + (size_t)r0 * j.w, j.src + (size_t)r0 * j.stride, (size_t)(r1 - r0) * row_bytes);
                } else {
                    for (int r = r0; r < r1; ++r) stage_copy(j.stage + (size_t)r * j.w, j.src + (size_t)r * j.stride, row_bytes);
                }
            }
            if (pieces_left[gen & 1u][k].fetch_sub(1, std::memory_order_acq_rel) != 1) return true;  // not the band's last piece
            bool ok = hipMemcpyAsync(j.dst + (size_t)b0 * j.w, j.stage + (size_t)b0 * j.w, (size_t)(b1 - b0) * row_bytes,
                                     hipMemcpyHostToDevice, j.up) == hipSuccess;
            if (ok && j.band_ev) ok = hipEventRecord(j.band_ev[k], j.up) == hipSuccess;
            if (!ok) {
                (void)hipGetLastError();
                failed.store(1);
            }
            if (j.ready) j.ready[k].store(ok ? 1 : -1, std::memory_order_release);
            bands_left.fetch_sub(1, std::memory_order_acq_rel);  // (last: the descriptor is not touched after this)
            return true;
        }
    }
    void work(uint32_t gen) {
        while (work_one(gen)) {
        }
    }
    // Helpers between jobs: a short spin (frames of a batch follow each other within microseconds), then they BLOCK on a
    // condition variable -- a library behind librectify.h must not keep eight threads polling in a process that is doing
    // nothing (round 3 did: 20 us naps for ever).  begin() wakes them only if somebody sleeps.
    std::mutex mu;
    std::condition_variable cv;
    std::atomic<int> sleepers{0};
    std::atomic<int> live{0};  // helpers that have a device and are taking jobs (the caller stages alone if none is)
    void start(lr_context* ctx, int helpers) {
        c = ctx;
        live.store(helpers, std::memory_order_relaxed);
        for (int t = 0; t < helpers; ++t)
            th.emplace_back([this]() {
                if (hipSetDevice(c->device) != hipSuccess) {  // a helper less; whoever waits for bands works on them itself
                    (void)hipGetLastError();
                    live.fetch_sub(1, std::memory_order_acq_rel);
                    return;
                }
                bind_this_thread_near(c->device);
                uint32_t last = 0;
                int spins = 0;
                while (!quit.load(std::memory_order_acquire)) {
                    const uint32_t g = job.load(std::memory_order_seq_cst);
                    if (g == last) {
                        if (++spins < 512) {
                            std::this_thread::yield();
                            continue;
                        }
                        std::unique_lock<std::mutex> lk(mu);
                        sleepers.fetch_add(1, std::memory_order_seq_cst);
                        cv.wait(lk, [&]() { return quit.load(std::memory_order_acquire) || job.load(std::memory_order_seq_cst) != last; });
                        sleepers.fetch_sub(1, std::memory_order_seq_cst);
                        spins = 0;
                        continue;
                    }
                    spins = 0;
                    last = g;
                    work(g);
                }
            });
    }
    void wake() {
        if (sleepers.load(std::memory_order_seq_cst) > 0) {
            { std::lock_guard<std::mutex> lk(mu); }
            cv.notify_all();
        }
    }
    // stages one frame (rows as in upload_rows) and enqueues its transfers; returns when every band is enqueued
    int run(float* dst_, float* stage_, const float* buffer, int w_, int h_, int stride_, hipStream_t up_,
            size_t band_bytes = (size_t)4 << 20, int pieces_ = 1) {
        const uint32_t g = begin(dst_, stage_, buffer, w_, h_, stride_, up_, nullptr, nullptr, band_bytes, pieces_);
        work(g);
        return finish();
    }
    // the two halves of run(): publish the job (the helpers start on it), and wait for its last band
    uint32_t begin(float* dst_, float* stage_, const float* buffer, int w_, int h_, int stride_, hipStream_t up_,
                   hipEvent_t* band_ev_, std::atomic<int>* ready_, size_t band_bytes = (size_t)4 << 20, int pieces_ = 0) {
        if (stride_ < 0) {
            buffer = buffer + (std::ptrdiff_t)(h_ - 1) * stride_;
            stride_ = -stride_;
        }
        const uint32_t g = job.load(std::memory_order_relaxed) + 1u;
        Job& j = jobs[g & 1u];
        j.dst = dst_;
        j.stage = stage_;
        j.src = buffer;
        j.w = w_;
        j.h = h_;
        j.stride = stride_;
        j.up = up_;
        j.band_ev = band_ev_;
        j.ready = ready_;
        j.rows_per_band = (int)std::max<size_t>(1, band_bytes / ((size_t)w_ * sizeof(float)));
        j.n_bands = (h_ + j.rows_per_band - 1) / j.rows_per_band;
        if (j.n_bands > kMaxBands) {  // (a frame of more than 1 GiB: fewer, larger bands)
            j.rows_per_band = (h_ + kMaxBands - 1) / kMaxBands;
            j.n_bands = (h_ + j.rows_per_band - 1) / j.rows_per_band;
        }
        j.pieces = pieces_ > 0 ? std::min(pieces_, kPieces) : (band_ev_ ? kPieces : 1);  // (band events = the single-frame path)
        for (int k = 0; k < j.n_bands; ++k) pieces_left[g & 1u][k].store(j.pieces, std::memory_order_relaxed);
        failed.store(0, std::memory_order_relaxed);  // (per frame: every band of the previous one has been accounted for)
        bands_left.store(j.n_bands, std::memory_order_relaxed);
        total.store(((uint64_t)g << 32) | (uint32_t)(j.n_bands * j.pieces), std::memory_order_release);
        next.store((uint64_t)g << 32, std::memory_order_release);
        job.store(g, std::memory_order_seq_cst);
        wake();
        return g;
    }
    int finish() {
        int spins = 0;
        while (bands_left.load(std::memory_order_acquire) > 0) {
            if (live.load(std::memory_order_acquire) == 0 && work_one(job.load(std::memory_order_relaxed))) continue;
            if (++spins < 256) std::this_thread::yield();
            else std::this_thread::sleep_for(std::chrono::microseconds(10));
        }
        return failed.load() ? 1 : 0;
    }
    ~StagingCrew() {
        quit.store(true, std::memory_order_seq_cst);
        { std::lock_guard<std::mutex> lk(mu); }
        cv.notify_all();
        for (auto& t : th) t.join();
    }
};

int ctx_upload_frame(lr_context* c, int slot, const float* buffer, int w, int h, int stride, int num_threads) {
    LR_HIP(hipSetDevice(c->device));
    if (w < 1 || h < 1 || buffer == nullptr) {
        set_error("upload: bad frame");
        return 1;
    }
    if ((stride < 0 ? -stride : stride) < w) {
        set_error("upload: |stride| smaller than the width");
        return 1;
    }
    if (ensure_copy_stream(c)) return 1;
    hipStream_t up = c->copy_stream;
    const size_t npix = (size_t)w * h;
    if (c->cap_slot[slot] < npix) {
        LR_HIP(hipStreamSynchronize(c->stream));
        LR_HIP(hipStreamSynchronize(up));
        if (dev_alloc(c->d_img_slot[slot], npix)) return 1;
        c->cap_slot[slot] = npix;
    }
    float* stage = nullptr;
    if (!is_page_locked(buffer)) {
        if (c->cap_stage[slot] < npix) {
            LR_HIP(hipStreamSynchronize(up));
            if (c->h_stage[slot]) (void)hipHostFree(c->h_stage[slot]);
            c->h_stage[slot] = nullptr;
            c->cap_stage[slot] = 0;
            LR_HIP(hipHostMalloc((void**)&c->h_stage[slot], npix * sizeof(float)));
            c->cap_stage[slot] = npix;
        }
        // the DMA that last read this staging buffer has long finished (its frame has been processed), but make sure
        LR_HIP(hipEventSynchronize(c->ev_up[slot]));
        stage = c->h_stage[slot];
    }
    if (upload_rows(c, c->d_img_slot[slot], stage, buffer, w, h, stride, num_threads, up)) return 1;
    LR_HIP(hipEventRecord(c->ev_up[slot], up));
    return 0;
}

int ctx_create(int device, lr_context** out) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        set_error("no HIP device available: librectify_amd has no CPU fallback");
        return 1;
    }
    if (device < 0 || device >= ndev) {
        set_error("device index out of range");
        return 1;
    }
    LR_HIP(hipSetDevice(device));
    lr_context* c = new lr_context();
    c->device = device;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        set_error("hipStreamCreate failed");
        return 1;
    }
    // Every allocation is checked: a create that ran out of memory must fail here, cleanly, and not hand out a context
    // whose first kernel faults on a null buffer (six lanes and their pools hold several GB -- exactly when it can happen).
    // LIBRECTIFY_TEST_CREATE_FAIL=n (test hook) makes the n-th allocation of a create report hipErrorOutOfMemory.
    {
        static const int fail_at = std::getenv("LIBRECTIFY_TEST_CREATE_FAIL") ? std::atoi(std::getenv("LIBRECTIFY_TEST_CREATE_FAIL")) : 0;
        int n_alloc = 0;
        hipError_t bad = hipSuccess;
        const char* what = "";
        auto check = [&](hipError_t e, const char* name) {
            if (++n_alloc == fail_at) e = hipErrorOutOfMemory;
            if (e != hipSuccess && bad == hipSuccess) {
                bad = e;
                what = name;
            }
        };
        for (auto& e : c->ev) check(hipEventCreate(&e), "hipEventCreate");
        check(hipEventCreateWithFlags(&c->ev_wait, hipEventBlockingSync | hipEventDisableTiming), "hipEventCreate");
        static const bool lane_debug = std::getenv("LIBRECTIFY_LANE_DEBUG") != nullptr;  // (its timeline times the uploads)
        for (auto& e : c->ev_up) check(hipEventCreateWithFlags(&e, lane_debug ? hipEventDefault : hipEventDisableTiming), "hipEventCreate");
        check(hipMalloc((void**)&c->maxmag, sizeof(float)), "hipMalloc(maxmag)");
        check(hipMalloc((void**)&c->d_counts, 64 * sizeof(uint32_t)), "hipMalloc(counts)");
        check(hipMalloc((void**)&c->d_gctl, kGcWords * sizeof(uint32_t)), "hipMalloc(peeling control block)");
        check(hipMalloc((void**)&c->d_gnorm, 4 * sizeof(float)), "hipMalloc(normalisation)");
        check(hipMalloc((void**)&c->d_models, 64 * sizeof(float)), "hipMalloc(models)");
        check(hipMalloc((void**)&c->d_best_slots, kRansacBestSlots * sizeof(unsigned long long)), "hipMalloc(best slots)");
        check(hipHostMalloc((void**)&c->h_counts, 128 * sizeof(uint32_t)), "hipHostMalloc(counts)");
        check(hipHostMalloc((void**)&c->h_best, 2 * sizeof(float)), "hipHostMalloc(best)");
        if (bad == hipSuccess) check(hipMemset(c->d_models, 0, 64 * sizeof(float)), "hipMemset(models)");
        if (bad == hipSuccess) check(hipMemset(c->d_counts, 0, 64 * sizeof(uint32_t)), "hipMemset(counts)");
        if (bad == hipSuccess) check(hipMemset(c->d_best_slots, 0, kRansacBestSlots * sizeof(unsigned long long)), "hipMemset(best slots)");
        if (bad != hipSuccess) {
            (void)hipGetLastError();
            const std::string msg = std::string("lr_context_create: ") + what + " failed: " + hipGetErrorString(bad);
            ctx_destroy(c);  // (frees whatever was made; null members are skipped)
            set_error(msg);
            return 1;
        }
    }
    init_constants(c);
    const char* env = std::getenv("LIBRECTIFY_SEED");
    c->ransac_seed = env ? std::strtoull(env, nullptr, 0) : 0ull;
    c->timing_on = std::getenv("LIBRECTIFY_STAGE_TIMES") != nullptr;
    if (const char* e = std::getenv("LIBRECTIFY_FLOOD_LOGS")) c->flood_logs = std::atoi(e) != 0;
    if (const char* e = std::getenv("LIBRECTIFY_FLOOD_MULTI")) c->flood_multi = std::atoi(e) != 0;  // (opt-in: DESIGN.md section 7, round 4)
    const char* fm = std::getenv("LIBRECTIFY_FLOOD_MODE");
    if (fm) c->flood_mode = std::atoi(fm);
    *out = c;
    return 0;
}

void ctx_destroy(lr_context* c) {
    if (!c) return;
    for (lr_context* wc : c->workers) ctx_destroy(wc);
    c->workers.clear();
    for (lr_context* pc : c->peers) ctx_destroy(pc);
    c->peers.clear();
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    void* ptrs[] = {c->d_img_slot[0], c->d_img_slot[1], c->dx, c->dy, c->dmask, c->cand, c->cand_count, c->tile_max, c->tile_pass, c->tile_off,
                    c->maxmag, c->keys_a, c->keys_b, c->d_counts, c->seed_idx, c->seed_bin, c->seed_thr, c->seed_size,
                    c->label, c->queue, c->comp_rank, c->comp_seed, c->comp_off, c->cursor, c->px_a, c->px_b,
                    c->scratch_w, c->d_lines, c->temp, c->d_model, c->d_best_slots,
                    c->fb.blocked, c->fb.count, c->fb.flags, c->fb.state, c->fb.tier, c->fb.blk, c->fb.act_a, c->fb.act_b,
                    c->fb.ctrl, c->fb.big_list, c->fb.handover, c->fb.waypoints, c->fb.multi_list, c->fb.log_off, c->fb.log_len, c->fb.log_buf, c->fb.dirty, c->fb.giant_mask, c->fb.slab_ring, c->fb.slab_hash, c->d_pairs, c->d_peak, c->d_weights,
                    c->d_samples, c->d_hcounts, c->comp_large, c->huge.tab, c->huge.jobs, c->huge.list, c->d_tables, c->d_orig, c->d_inl, c->d_flines, c->d_gctl,
                    c->d_gnorm, c->d_models, c->d_refine_table, c->d_refine_edges, c->d_cht_acc, c->d_cht_idx, c->d_cht_peak, c->d_rec, c->d_recflags};
    for (void* p : ptrs)
        if (p) (void)hipFree(p);
    if (c->flood_aux) {
        (void)hipStreamSynchronize(c->flood_aux);
        (void)hipStreamDestroy(c->flood_aux);
    }
    for (auto& e : c->flood_fork)
        if (e) (void)hipEventDestroy(e);
    for (auto& e : c->flood_join)
        if (e) (void)hipEventDestroy(e);
    delete static_cast<StagingCrew*>(c->crew);
    c->crew = nullptr;
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    for (auto& e : c->band_ev)
        if (e) (void)hipEventDestroy(e);
    for (float* p : c->h_stage)
        if (p) (void)hipHostFree(p);
    for (auto& e : c->ev_up)
        if (e) (void)hipEventDestroy(e);
    if (c->ev_wait) (void)hipEventDestroy(c->ev_wait);
    for (float* p : c->ring_img)
        if (p) (void)hipFree(p);
    for (float* p : c->ring_stage)
        if (p) (void)hipHostFree(p);
    for (auto& e : c->ring_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->h_res) (void)hipHostFree(c->h_res);
    if (c->h_model) (void)hipHostFree(c->h_model);
    if (c->h_counts) (void)hipHostFree(c->h_counts);
    if (c->h_best) (void)hipHostFree(c->h_best);
    if (c->h_pairs) (void)hipHostFree(c->h_pairs);
    if (c->h_weights) (void)hipHostFree(c->h_weights);
    if (c->h_samples) (void)hipHostFree(c->h_samples);
    if (c->h_hcounts) (void)hipHostFree(c->h_hcounts);
    if (c->h_recflags) (void)hipHostFree(c->h_recflags);
    if (c->h_rec) (void)hipHostFree(c->h_rec);
    for (auto& e : c->prosac_ev)
        if (e) (void)hipEventDestroy(e);
    if (c->h_cht_idx) (void)hipHostFree(c->h_cht_idx);
    if (c->h_cht_peak) (void)hipHostFree(c->h_cht_peak);
    for (auto& e : c->ev)
        if (e) (void)hipEventDestroy(e);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

// ---- stages ------------------------------------------------------------------------------
//
// Every stage is an "enqueue" function that launches its kernels on the context stream and never waits: seed,
// component, pixel and line counts stay in device memory (d_counts, d_gctl) and the kernels read them there;
// launches cover capacities (seed_cap, line_cap) chosen before the counts exist.  The frame driver (run_frame) enqueues
// all stages back to back and synchronises once; the staged API (lr_stage_*) wraps one enqueue function each and waits
// for it, because its callers want the counts.

namespace {

// d_counts words
enum { kCntSeeds = 0, kCntComp = 1, kCntPx = 2, kCntLarge = 4 };

uint32_t round_up(uint32_t v, uint32_t m) { return (v + m - 1) / m * m; }

// capacity of the seed sort for a frame of this size when nothing better is known: one seed per 32 pixels (the
// synthetic 4K frame has one per 204, the reference's doc image one per 85)
uint32_t initial_seed_cap(size_t npix) {
    return (uint32_t)std::min<size_t>(npix, std::max<size_t>(round_up((uint32_t)(npix / 32), 1024), 4096));
}

uint32_t line_cap_for(const lr_context* c) {
    return (uint32_t)std::min<size_t>(c->seed_cap, c->cap_pix / 6 + 16);
}

int ensure_group_capacity(lr_context* c, size_t n_lines, size_t n_iter) {
    if (n_lines > c->cap_glines) {
        LR_HIP(hipStreamSynchronize(c->stream));
        const size_t cl = std::max<size_t>(n_lines, 4096);
        if (dev_alloc(c->d_tables, 3 * 8 * cl) || dev_alloc(c->d_orig, 3 * cl) || dev_alloc(c->d_inl, 4 * cl)) return 1;
        c->cap_glines = cl;
    }
    if (n_lines > c->cap_flines) {
        LR_HIP(hipStreamSynchronize(c->stream));
        if (dev_alloc(c->d_flines, n_lines + 16)) return 1;
        c->cap_flines = n_lines + 16;
    }
    (void)n_iter;  // (scores are not stored any more: kernels_ransac.hip)
    return 0;
}

PencilTable table_of(lr_context* c, int which) {
    const size_t cl = c->cap_glines;
    float* b = c->d_tables + (size_t)which * 8 * cl;
    return PencilTable{b, b + cl, b + 2 * cl, b + 3 * cl, b + 4 * cl, b + 5 * cl, b + 6 * cl, b + 7 * cl,
                       c->d_orig + (size_t)which * cl};
}

int ensure_result_block(lr_context* c, size_t lines) {
    if (lines <= c->res_lines_cap && c->h_res) return 0;
    LR_HIP(hipStreamSynchronize(c->stream));
    if (c->h_res) (void)hipHostFree(c->h_res);
    c->h_res = nullptr;
    LR_HIP(hipHostMalloc((void**)&c->h_res, kResHeaderBytes + lines * sizeof(LineSegment)));
    c->res_lines_cap = lines;
    return 0;
}

FloodBuffers flood_buffers_for(lr_context* c) {
    FloodBuffers fbuf = c->fb;
    // test hooks: 2 and 3 exercise the slab and exhausted-storage paths (no second LDS tier, no / two slabs),
    // 4 the slab path with the full pool, 5 a stall during the hold-back, 6 / 7 the second tier's team running out of storage
    if (c->flood_mode >= 2 && c->flood_mode <= 4) fbuf.second_tier = false;
    fbuf.second_tier_from_start = c->flood_big_hint;
    fbuf.hold_from_start = c->flood_hold_hint;
    fbuf.staged_from_start = c->flood_staged_hint;
    if (c->flood_mode == 2) fbuf.n_slabs = 0;
    if (c->flood_mode == 3) fbuf.n_slabs = 2;
    if (c->flood_mode == 5) {  // second tier with room for one seed per round and no slab: the rounds stall while
        fbuf.n_slabs = 0;      // the weakest seeds are held back, and must still hand over to the ordered tail
        fbuf.big_cap_override = 1;
    }
    if (c->flood_mode == 6) fbuf.team_tile_cap = 200;  // the second tier's team runs out early: the whole team moves into a
    if (c->flood_mode == 7) {                           // global slab and goes on there (6), or there is none: incomplete
        fbuf.team_tile_cap = 200;                       // walk, barrier, ordered tail (7)
        fbuf.n_slabs = 0;
    }
    static const bool partial_off = std::getenv("LIBRECTIFY_FLOOD_PARTIAL") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_PARTIAL")) == 0;
    fbuf.partial_commits = !partial_off && c->flood_partial;
    fbuf.multi_source = c->flood_multi;
    fbuf.rewalk_logs = c->flood_logs && !c->flood_multi;  // (way-points, when asked for, instead)
    fbuf.log_sweep = c->flood_log_sweep;
    fbuf.log_from_round = c->flood_log_from;
    fbuf.log_min_tiles = c->flood_log_min;
    fbuf.log_walk_tiles = c->flood_log_walk;
    static const bool giants_off = std::getenv("LIBRECTIFY_FLOOD_GIANTS") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_GIANTS")) == 0;
    fbuf.giant_hold = c->flood_mode == 1 && !giants_off;  // (the storage test hooks -- modes 2-7 -- keep their slabs)
    static const bool giant_step_off = std::getenv("LIBRECTIFY_FLOOD_GIANT_STEP") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_GIANT_STEP")) == 0;
    fbuf.giant_step = !giant_step_off && c->flood_giant_step;
    fbuf.giant_parent = reinterpret_cast<uint32_t*>(c->queue);
    fbuf.rewalk_big = c->flood_logbig_hint && !c->flood_logbig_off;  // (the context's last frame had walks beyond the first tier)
    static const int aux_env = std::getenv("LIBRECTIFY_FLOOD_MULTI_BESIDE") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_MULTI_BESIDE")) : -1;  // (experiment: 0 = after the exploration, N = beside it in rounds 2 .. N + 1)
    if (c->flood_aux && c->flood_aux_on && aux_env != 0 && c->flood_fork.size() == c->flood_join.size()) {
        fbuf.aux_stream = c->flood_aux;
        fbuf.fork_events = c->flood_fork.data();
        fbuf.join_events = c->flood_join.data();
        fbuf.n_fork_events = (int)c->flood_fork.size();
        if (aux_env > 0) fbuf.multi_round_last = aux_env;
    }
    if (c->flood_staged) fbuf.win_first_shift = 3;
    fbuf.blind_rounds = c->flood_rounds_hint;
    // rounds just in time (FloodBuffers::host_progress): what the last frame needed less one at once (three on a new context)
    static const bool jit_off = std::getenv("LIBRECTIFY_FLOOD_JIT") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_JIT")) == 0;
    static_assert(16 + kFloodCtrlWords <= 72, "the control block's copy ends where the report word begins");
    fbuf.host_progress = c->h_counts + 72;
    fbuf.host_ctrl = c->h_counts + 16;
    // (LIBRECTIFY_FLOOD_CALM_HINT=0: every blind round brings the second tier's launch, as until round 5)
    static const int calm_hint_env = std::getenv("LIBRECTIFY_FLOOD_CALM_HINT") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_CALM_HINT")) : 1;
    fbuf.calm_hint = calm_hint_env != 0 && c->flood_calm_hint;
    fbuf.jit_sleep_us = c->flood_jit_sleep_us;
    // (at most four rounds blindly -- the rounds that always bring their `rest` launch, kernels_flood.hip kRestRounds: a later
    // blind round whose list is longer than its grid walks only a part of it, and lists stay long while a window is closed in
    // front of waiting seeds (a frame of soft blobs went to the ordered tail that way now and then: 42 -> 200 ms); a round
    // enqueued just in time knows its list's length and brings the launch when it needs it.  LIBRECTIFY_FLOOD_JIT_FIRST_MAX)
    static const int jit_first_max = std::getenv("LIBRECTIFY_FLOOD_JIT_FIRST_MAX") ? std::max(1, std::atoi(std::getenv("LIBRECTIFY_FLOOD_JIT_FIRST_MAX"))) : 4;
    fbuf.jit_first = (c->flood_jit && !jit_off) ? std::min(c->flood_rounds_last > 0 ? std::max(c->flood_rounds_last - 1, 2) : 3, jit_first_max) : 0;
    // (the lanes of a batch: LIBRECTIFY_FLOOD_JIT_FIRST_LANES / _LEAD_LANES = rounds enqueued blindly at most / rounds kept ahead)
    static const int first_lanes = std::getenv("LIBRECTIFY_FLOOD_JIT_FIRST_LANES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_JIT_FIRST_LANES")) : 0;
    static const int lead_lanes = std::getenv("LIBRECTIFY_FLOOD_JIT_LEAD_LANES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_JIT_LEAD_LANES")) : 0;
    static const int lead_single = std::getenv("LIBRECTIFY_FLOOD_JIT_LEAD") ? std::max(0, std::atoi(std::getenv("LIBRECTIFY_FLOOD_JIT_LEAD"))) : 0;
    fbuf.jit_lead = lead_single;
    if (c->flood_jit_sleep_us > 0 && fbuf.jit_first > 0) {
        if (first_lanes > 0) fbuf.jit_first = std::min(fbuf.jit_first, first_lanes);
        fbuf.jit_lead = std::max(lead_lanes, 0);
    }
    return fbuf;
}

FloodFrame flood_frame_for(lr_context* c) {
    return FloodFrame{c->dx, c->dy, c->dmask, c->w, c->h, c->seed_idx, c->seed_bin, c->seed_thr,
                      c->d_counts + kCntSeeds, c->seed_cap, c->trig, c->label, c->seed_size, c->queue};
}

// everything of a frame's first stage but the launch: workspace, seed-sort capacity, state of the last run
int prepare_frame(lr_context* c, int w, int h) {
    LR_HIP(hipSetDevice(c->device));
    if (w < 5 || h < 5) {
        set_error("image smaller than the 5x5 filter");
        return 1;
    }
    if (ctx_ensure_image_capacity(c, w, h)) return 1;
    if (c->w != w || c->h != h || c->seed_cap == 0) c->seed_cap = initial_seed_cap((size_t)w * h);
    if (c->seed_cap_once) {  // test hook (lr_set_seed_capacity): this frame starts with the given capacity
        c->seed_cap = (uint32_t)std::min<size_t>((size_t)w * h, c->seed_cap_once);
        c->seed_cap_once = 0;
    }
    c->w = w;
    c->h = h;
    c->n_seeds = c->n_comp = c->n_px = 0;
    c->dmask_consumed = false;
    for (bool& v : c->stage_valid) v = false;
    return 0;
}

int enqueue_filter(lr_context* c, const float* d_image, int w, int h, int stride) {
    if (prepare_frame(c, w, h)) return 1;
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[0], c->stream));
    if (launch_filter(d_image, w, h, stride, c->fconsts, c->dx, c->dy, c->dmask, c->cand, c->cand_count, c->tile_max,
                      c->stream))
        return 1;
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[1], c->stream));
    return 0;
}

int enqueue_seeds(lr_context* c) {
    const FilterGeom fg = filter_geometry(c->w, c->h);
    if (launch_seed_select(c->cand, c->cand_count, c->tile_max, fg.n_tiles, fg.cand_cap, c->seed_keep_ratio, c->maxmag,
                           c->tile_pass, c->tile_off, c->keys_a, c->seed_cap, c->d_counts + kCntSeeds, ++c->select_tag ? c->select_tag : ++c->select_tag,
                           c->stream))
        return 1;
    if (launch_seed_order(c->keys_a, c->keys_b, c->d_counts + kCntSeeds, c->seed_cap, c->dx, c->dy, c->trig, kTraceTolerance,
                          c->seed_idx, c->seed_bin, c->seed_thr, c->stream))
        return 1;
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[2], c->stream));
    return 0;
}

// the flood's rounds, blindly (parallel modes) or the single-wave ordered kernel (mode 0)
int enqueue_flood(lr_context* c) {
    const size_t npix = (size_t)c->w * c->h;
    c->flood_rounds = 1;
    if (c->flood_mode == 0) {
        if (launch_label_init(c->label, npix, c->stream)) return 1;  // (the parallel rounds' set-up kernel does it itself)
        if (launch_flood_ordered(c->dx, c->dy, c->dmask, c->w, c->h, c->seed_idx, c->seed_bin, c->seed_thr,
                                 c->d_counts + kCntSeeds, c->seed_cap, c->trig, c->label, c->seed_size, c->queue, c->stream))
            return 1;
    } else {
        if (ensure_flood_buffers(c)) return 1;
        if (flood_enqueue(flood_buffers_for(c), flood_frame_for(c), &c->flood_prog, c->h_counts + 16, c->stream)) return 1;
        // the commit pass clears the direction mask of every labelled pixel (kernels_flood.hip): the filter
        // output is consumed, a second flood needs lr_stage_filter + lr_stage_seeds again
        c->stage_valid[0] = c->stage_valid[1] = false;
        c->dmask_consumed = true;
    }
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[3], c->stream));
    return 0;
}

// After a synchronisation: did the blind rounds finish the flood?  If not, finish it now (synchronises).
int finish_flood(lr_context* c, bool* extra) {
    *extra = false;
    if (c->flood_mode == 0) return 0;
    if (flood_finish(flood_buffers_for(c), flood_frame_for(c), &c->flood_prog, c->h_counts + 16, &c->flood_rounds,
                     c->flood_tiers, extra, c->stream))
        return 1;
    static const bool call_debug = std::getenv("LIBRECTIFY_CALL_DEBUG") != nullptr;
    if (call_debug)
        std::fprintf(stderr, "flood: %d rounds, %u walks in the second tier, %u of them long, hold-back phase %u (started with it: %d)\n",
                     c->flood_rounds, c->flood_tiers[0], c->flood_tiers[8], c->flood_tiers[3], (int)c->flood_hold_hint);
    c->flood_logbig_hint = c->flood_tiers[0] > 0;
    c->flood_calm_hint = c->flood_tiers[0] == 0 && c->flood_tiers[1] == 0 && c->flood_tiers[15] == 0;
    static const int hints_env = std::getenv("LIBRECTIFY_FLOOD_HINTS") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_HINTS")) : 1;  // (experiment knob)
    if (hints_env == 0) {  // as in round 2
        c->flood_big_hint = c->flood_tiers[0] > 0 || c->flood_tiers[1] > 0;
        c->flood_hold_hint = c->flood_tiers[3] != 0;
    } else {
        // The second tier is always there: a frame of regions that follows a frame of lines on this context used to run its
        // first batch of rounds without it, every long walk in a global slab (6.4 instead of 1.5 ms of flood on the natural
        // 4K frame), and an empty launch of its kernel costs a round 5 us.  The hold-back starts with the frame only after a
        // frame of REGIONS (many walks beyond the first tier's table): engaged from the start on a frame of lines it costs
        // three rounds (1.55 instead of 1.10 ms), and the old rule -- "the last frame engaged it" -- kept itself alive from
        // frame to frame once a single frame had.
        c->flood_big_hint = true;
        c->flood_hold_hint = c->flood_tiers[3] != 0 && c->flood_tiers[8] >= 16;
        // A frame that went on staged (many walks held back in its first round) hands that on; a frame that STARTED staged
        // keeps handing it on while its floods still look like regions (walks in the second tier: the staged start itself
        // keeps the giants away, so their count says nothing any more).
        static const int staged_keep = std::getenv("LIBRECTIFY_FLOOD_STAGED_KEEP") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_STAGED_KEEP")) : 64;
        c->flood_staged_hint = c->flood_tiers[14] != 0 && (!c->flood_staged_hint || (int)c->flood_tiers[0] >= staged_keep);
        // (a frame that started staged cannot tell whether it would have gone staged by itself: every sixteenth frame of such
        // a run starts without the hint and finds out -- a ramp under noise pays 1.6 ms for that frame, soft blobs that
        // inherited the hint from other content are rid of it)
        c->flood_staged_streak = c->flood_staged_hint ? c->flood_staged_streak + 1 : 0;
        if (c->flood_staged_streak >= 16) {
            c->flood_staged_hint = false;
            c->flood_staged_streak = 0;
        }
        if (c->flood_staged_hint) c->flood_hold_hint = false;
        // (The verdict "many long walks" -- early hand-over to the second tier, flood_advance -- is NOT carried over: started
        // with it, the natural 4K frame sends 735 walks to the second tier in round one and its flood takes 1.88 ms instead
        // of 1.5, and a frame of lines that follows pays 0.7 ms for the wrong guess.)
    }

    // blind rounds of the next frame: what this one needed plus two, decaying slowly
    static const int blind_extra = std::getenv("LIBRECTIFY_BLIND_EXTRA") ? std::atoi(std::getenv("LIBRECTIFY_BLIND_EXTRA")) : 2;  // (experiment knob)
    static const int blind_min = std::getenv("LIBRECTIFY_BLIND_MIN") ? std::atoi(std::getenv("LIBRECTIFY_BLIND_MIN")) : 6;
    c->flood_rounds_hint = std::max(std::max(c->flood_rounds + blind_extra, blind_min), c->flood_rounds_hint - 1);
    c->flood_rounds_last = c->flood_rounds;
    return 0;
}

int enqueue_fit(lr_context* c) {
    const size_t npix = (size_t)c->w * c->h;
    const uint32_t comp_cap = line_cap_for(c);
    // Floods of more than 2^14 pixels have launches of their own (kernels_fit.hip: huge_count_kernel, huge_sort_kernel), left
    // out when the host KNOWS that the frame has none: the flood's rounds, enqueued just in time, report their largest commit.
    const bool with_huge = !(c->flood_mode != 0 && c->flood_prog.sizes_known && c->flood_prog.max_flood <= (1u << 14));
    if (launch_component_offsets(c->seed_size, c->d_counts + kCntSeeds, c->seed_cap, kComponentMinSize, c->comp_rank,
                                 c->comp_seed, c->comp_off, c->d_counts + kCntComp, c->comp_large,
                                 (uint32_t)(c->cap_pix / 64 + 16), c->d_counts + kCntLarge, c->temp, c->temp_bytes, ++c->fit_tag ? c->fit_tag : ++c->fit_tag, c->cursor,
                                 with_huge ? c->huge : HugeSort{}, c->stream))
        return 1;
    if (launch_component_scatter(c->label, npix, c->comp_rank, c->comp_off, c->cursor, c->px_a, c->huge, c->d_counts + kCntLarge, with_huge,
                                 c->stream))
        return 1;
    // (the sorted seed keys are dead once the seeds are set up: their buffer is the large lists' sorting scratch)
    if (launch_component_sort(c->px_a, c->px_b, c->comp_off, c->d_counts + kCntComp, comp_cap, c->comp_large,
                              (uint32_t)(c->cap_pix / 64 + 16), c->d_counts + kCntLarge, reinterpret_cast<uint32_t*>(c->keys_b),
                              c->cursor, c->huge, with_huge, c->stream))
        return 1;
    static const bool fit_debug = std::getenv("LIBRECTIFY_FIT_DEBUG") != nullptr;
    if (fit_debug) {
        uint32_t cnt[16];
        LR_HIP(hipStreamSynchronize(c->stream));
        LR_HIP(hipMemcpy(cnt, c->d_counts, sizeof(cnt), hipMemcpyDeviceToHost));
        std::fprintf(stderr, "fit: with_huge %d (sizes known %d, largest flood %u); seeds %u comps %u px %u; lists: %u of 65..1024 px, %u longer, %u huge in %u buckets; counter %u\n",
                     (int)with_huge, (int)c->flood_prog.sizes_known, c->flood_prog.max_flood, cnt[0], cnt[1], cnt[2], cnt[4], cnt[5], cnt[6], cnt[7], cnt[10]);
        const uint32_t nj = std::min<uint32_t>(cnt[7], 6);
        std::vector<uint32_t> jobs(4 * nj + 4);
        if (nj) LR_HIP(hipMemcpy(jobs.data(), c->huge.jobs, nj * 16, hipMemcpyDeviceToHost));
        for (uint32_t j = 0; j < nj; ++j) std::fprintf(stderr, "   bucket job %u: start %u, %u px, table word %u\n", j, jobs[4 * j], jobs[4 * j + 1], jobs[4 * j + 2]);
    }
    if (launch_fit(c->px_b, c->px_a, c->comp_off, c->comp_seed, c->d_counts + kCntComp, comp_cap, c->seed_bin, c->dx, c->dy, c->w,
                   c->trig, c->scratch_w, c->d_lines, c->cursor, c->huge, c->d_counts + kCntLarge, with_huge, c->stream))
        return 1;
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[4], c->stream));
    return 0;
}

// estimate_line_pencils (line_pencil.cpp:148-177) on the lines in d_flines, whose count, bounding box and control
// words a filter_lines / lines_bbox launch has left in d_gctl / d_gnorm.
int enqueue_groups(lr_context* c, uint32_t line_cap, int max_models, float inlier_deg, float garbage_deg, int n_iter,
                   uint64_t seed, bool model_done = false, bool gather = false) {
    if (max_models > kMaxPeelModels) {  // the refit models and the diagnostics slots of a frame are sized for this many
        set_error("estimate_line_pencils: max_models above " + std::to_string(kMaxPeelModels) + " (the reference uses 4, config.h:25)");
        return 1;
    }
    const float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    const PencilTable all = table_of(c, 0);
    PencilTable tab[2] = {table_of(c, 1), table_of(c, 2)};
    // (model_done: the launch that filtered the lines has written both tables already -- launch_filter_lines)
    if (!model_done && launch_pencil_model(c->d_flines, c->d_gctl, c->d_gnorm, all, tab[0], line_cap, c->stream)) return 1;
    const float degeneracy_tol = 0.05f;  // line_pencil.h:25
    for (int k = 0; k < max_models; ++k) {
        if (n_iter > 0 &&
            launch_ransac_score_dev(tab[k & 1].soa(), c->d_gctl, max_models, tol, degeneracy_tol, (uint32_t)n_iter, seed,
                                    c->d_best_slots, c->stream))
            return 1;
        // (gather: the last round's launch copies the frame's results into the page-locked block as well)
        const bool last = gather && k == max_models - 1;
        if (launch_peel(tab[k & 1], tab[(k + 1) & 1], all, c->d_best_slots, seed, tol, garbage_tol,
                        max_models, c->d_gctl, c->d_inl, c->d_flines, c->d_models, c->d_counts,
                        (uint32_t)std::min<size_t>(c->res_lines_cap, c->cap_flines), last ? c->h_res : nullptr, c->stream))
            return 1;
    }
    return 0;
}

void record_stage_times(lr_context* c, bool with_groups) {
    if (!c->timing_on) return;
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[1]);
    c->stage_ms[LR_T_FILTER] = ms;
    c->stage_ms[LR_T_FILTER_KERNEL] = ms;
    (void)hipEventElapsedTime(&ms, c->ev[1], c->ev[2]);
    c->stage_ms[LR_T_SEEDS] = ms;
    (void)hipEventElapsedTime(&ms, c->ev[2], c->ev[3]);
    c->stage_ms[LR_T_FLOOD] = ms;
    (void)hipEventElapsedTime(&ms, c->ev[3], c->ev[4]);
    c->stage_ms[LR_T_FIT] = ms;
    if (with_groups) {
        (void)hipEventElapsedTime(&ms, c->ev[5], c->ev[6]);
        c->stage_ms[LR_T_RANSAC] = ms;
        (void)hipEventElapsedTime(&ms, c->ev[0], c->ev[6]);
        c->stage_ms[LR_T_TOTAL] = ms;
    }
}

// The seed sort's capacity follows the frames: kept between 1.25 and 3 times the last frame's seed count (launches of
// the flood and the fit are sized by it, and empty workgroups are not free), grown at once when a frame overflows it
// (that frame is repeated).
void adapt_seed_cap(lr_context* c, uint32_t n_seeds) {
    const size_t npix = (size_t)c->w * c->h;
    if (n_seeds > c->seed_cap)
        c->seed_cap = (uint32_t)std::min<size_t>(npix, std::max<uint32_t>(4096, round_up(n_seeds * 2u, 1024)));
    else if (n_seeds < c->seed_cap / 3 || n_seeds > c->seed_cap / 5 * 4)
        c->seed_cap = (uint32_t)std::min<size_t>(npix, std::max<uint32_t>(4096, round_up(n_seeds + n_seeds / 2, 1024)));
}

}  // namespace

// (the staged API exists for tests and measurements: its stages are always timed)
struct TimingOn {
    lr_context* c;
    bool was;
    explicit TimingOn(lr_context* ctx) : c(ctx), was(ctx->timing_on) { c->timing_on = true; }
    ~TimingOn() { c->timing_on = was; }
};

int ctx_stage_filter(lr_context* c, const float* d_image, int w, int h, int stride) {
    TimingOn t(c);
    if (enqueue_filter(c, d_image, w, h, stride)) return 1;
    c->stage_valid[0] = true;
    return 0;
}

int ctx_stage_seeds(lr_context* c) {
    if (!c->stage_valid[0]) {
        set_error("lr_stage_seeds: run lr_stage_filter first");
        return 1;
    }
    TimingOn t(c);
    for (int attempt = 0;; ++attempt) {
        if (enqueue_seeds(c)) return 1;
        LR_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        LR_HIP(hipStreamSynchronize(c->stream));
        c->n_seeds = c->h_counts[0];
        if (c->n_seeds <= c->seed_cap) break;
        if (attempt > 0) {
            set_error("seed count exceeds the sort capacity twice");
            return 1;
        }
        adapt_seed_cap(c, c->n_seeds);  // more seeds than the sort was sized for: again with room
    }
    c->stage_valid[1] = true;
    return 0;
}

int ctx_stage_flood(lr_context* c) {
    if (!c->stage_valid[1]) {
        set_error(c->dmask_consumed ? "lr_stage_flood: the previous flood consumed the filter output; run lr_stage_filter and lr_stage_seeds again"
                                    : "lr_stage_flood: run lr_stage_seeds first");
        return 1;
    }
    TimingOn t(c);
    if (enqueue_flood(c)) return 1;
    LR_HIP(hipStreamSynchronize(c->stream));
    bool extra;
    if (finish_flood(c, &extra)) return 1;
    c->stage_valid[2] = true;
    return 0;
}

int ctx_stage_fit(lr_context* c, std::vector<LineSegment>& out) {
    if (!c->stage_valid[2]) {
        set_error("lr_stage_fit: run lr_stage_flood first");
        return 1;
    }
    out.clear();
    TimingOn t(c);
    if (enqueue_fit(c)) return 1;
    LR_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    LR_HIP(hipStreamSynchronize(c->stream));
    c->n_comp = c->h_counts[kCntComp];
    c->n_px = c->h_counts[kCntPx];
    if (c->n_comp > 0) {
        out.resize(c->n_comp);
        LR_HIP(hipMemcpyAsync(out.data(), c->d_lines, (size_t)c->n_comp * sizeof(LineSegment), hipMemcpyDeviceToHost,
                              c->stream));
        LR_HIP(hipStreamSynchronize(c->stream));
    }
    c->stage_valid[3] = true;
    record_stage_times(c, false);
    return 0;
}

int ctx_detect(lr_context* c, const float* d_image, int w, int h, int stride, std::vector<LineSegment>& raw) {
    if (ctx_stage_filter(c, d_image, w, h, stride)) return 1;
    if (ctx_stage_seeds(c)) return 1;
    if (ctx_stage_flood(c)) return 1;
    return ctx_stage_fit(c, raw);
}

// ---- RANSAC ------------------------------------------------------------------------------

int ctx_ransac_best(lr_context* c, const PencilModel& model, const std::vector<int>& indices, float tol, int n_iter,
                    uint64_t seed, uint32_t round, Vec3* best_h, float* best_score, int* best_iter) {
    LR_HIP(hipSetDevice(c->device));
    const size_t n = indices.size();
    *best_h = {0.f, 0.f, 0.f};  // the reference leaves best_h uninitialised when nothing scores (estimator.h:39)
    *best_score = 0.f;
    *best_iter = -1;
    if (n < 2 || n_iter <= 0) return 0;
    if (ctx_ensure_ransac_capacity(c, n, (size_t)n_iter)) return 1;
    float* hm = c->h_model;
    for (size_t j = 0; j < n; ++j) {
        const int i = indices[j];
        hm[0 * n + j] = model.anchor[i].x;
        hm[1 * n + j] = model.anchor[i].y;
        hm[2 * n + j] = model.direction[i].x;
        hm[3 * n + j] = model.direction[i].y;
        hm[4 * n + j] = model.length[i];
        hm[5 * n + j] = model.h[i].x;
        hm[6 * n + j] = model.h[i].y;
        hm[7 * n + j] = model.h[i].z;
    }
    LR_HIP(hipMemcpyAsync(c->d_model, hm, 8 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    PencilSoA m{c->d_model + 0 * n, c->d_model + 1 * n, c->d_model + 2 * n, c->d_model + 3 * n,
                c->d_model + 4 * n, c->d_model + 5 * n, c->d_model + 6 * n, c->d_model + 7 * n};
    // scoring launch + read-out: (score bits, iteration) arrive in the page-locked pair, no copy commands
    if (launch_ransac_score(m, (uint32_t)n, tol, model.degeneracy_tol, (uint32_t)n_iter, seed, round, c->d_best_slots,
                            reinterpret_cast<uint32_t*>(c->h_best), c->stream))
        return 1;
    LR_HIP(hipStreamSynchronize(c->stream));
    int32_t it;
    std::memcpy(&it, &c->h_best[1], sizeof(it));
    *best_score = c->h_best[0];
    *best_iter = it;
    if (it >= 0) {
        uint32_t a, b;
        sample_pair(seed, round, (uint32_t)it, (uint32_t)n, a, b);
        *best_h = model.fit(indices[a], indices[b]);
    }
    return 0;
}

// estimate_line_pencils (line_pencil.cpp:148-177) for lines the host holds: upload, peel on the device, download.
int ctx_estimate_line_pencils(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                              float garbage_deg, int n_iter, uint64_t seed) {
    if (lines.empty()) return 0;
    LR_HIP(hipSetDevice(c->device));
    const size_t n = lines.size();
    if (ensure_group_capacity(c, n, (size_t)std::max(n_iter, 1))) return 1;
    LR_HIP(hipMemcpyAsync(c->d_flines, lines.data(), n * sizeof(LineSegment), hipMemcpyHostToDevice, c->stream));
    if (launch_lines_bbox(c->d_flines, (uint32_t)n, c->d_gctl, c->d_gnorm, c->stream)) return 1;
    if (enqueue_groups(c, (uint32_t)n, max_models, inlier_deg, garbage_deg, n_iter, seed)) return 1;
    LR_HIP(hipMemcpyAsync(lines.data(), c->d_flines, n * sizeof(LineSegment), hipMemcpyDeviceToHost, c->stream));
    LR_HIP(hipStreamSynchronize(c->stream));
    return 0;
}

// ---- PROSAC (opt-in; reference prosac.h, never instantiated there) ---------------------------------

namespace {

int upload_model(lr_context* c, const PencilModel& model, const std::vector<int>& order, PencilSoA* out) {
    const size_t n = order.size();
    if (ctx_ensure_ransac_capacity(c, n, 1)) return 1;
    float* hm = c->h_model;
    for (size_t j = 0; j < n; ++j) {
        const int i = order[j];
        hm[0 * n + j] = model.anchor[i].x;
        hm[1 * n + j] = model.anchor[i].y;
        hm[2 * n + j] = model.direction[i].x;
        hm[3 * n + j] = model.direction[i].y;
        hm[4 * n + j] = model.length[i];
        hm[5 * n + j] = model.h[i].x;
        hm[6 * n + j] = model.h[i].y;
        hm[7 * n + j] = model.h[i].z;
    }
    LR_HIP(hipMemcpyAsync(c->d_model, hm, 8 * n * sizeof(float), hipMemcpyHostToDevice, c->stream));
    *out = PencilSoA{c->d_model + 0 * n, c->d_model + 1 * n, c->d_model + 2 * n, c->d_model + 3 * n,
                     c->d_model + 4 * n, c->d_model + 5 * n, c->d_model + 6 * n, c->d_model + 7 * n};
    return 0;
}

constexpr uint32_t kProsacRecCap = 32;  // new-best iterations of a chunk whose inlier flags come back with its counts

int ensure_prosac_buffers(lr_context* c, size_t n_lines, size_t n_pairs, size_t chunk) {
    if (n_pairs > c->cap_pairs) {
        LR_HIP(hipStreamSynchronize(c->stream));
        if (dev_alloc(c->d_pairs, 2 * n_pairs)) return 1;
        if (c->h_pairs) (void)hipHostFree(c->h_pairs);
        LR_HIP(hipHostMalloc((void**)&c->h_pairs, 2 * n_pairs * sizeof(int32_t)));
        if (!c->d_peak && dev_alloc(c->d_peak, 4)) return 1;
        c->cap_pairs = n_pairs;
    }
    if (n_lines > c->cap_wlines) {
        LR_HIP(hipStreamSynchronize(c->stream));
        const size_t cl = std::max<size_t>(n_lines, 4096);
        if (dev_alloc(c->d_weights, cl)) return 1;
        if (c->h_weights) (void)hipHostFree(c->h_weights);
        LR_HIP(hipHostMalloc((void**)&c->h_weights, cl * sizeof(float)));
        c->cap_wlines = cl;
    }
    // (two of everything a chunk of hypotheses uses: the next chunk is on the GPU while the host goes through the last one)
    if (n_lines * kProsacRecCap > c->cap_recflags) {
        LR_HIP(hipStreamSynchronize(c->stream));
        const size_t bytes = std::max<size_t>(n_lines, 4096) * kProsacRecCap;
        if (dev_alloc(c->d_recflags, 2 * bytes) || dev_alloc(c->d_rec, 2 * (kProsacRecCap + 1))) return 1;
        if (c->h_recflags) (void)hipHostFree(c->h_recflags);
        if (c->h_rec) (void)hipHostFree(c->h_rec);
        c->h_recflags = nullptr;
        c->h_rec = nullptr;
        LR_HIP(hipHostMalloc((void**)&c->h_recflags, 2 * bytes));
        LR_HIP(hipHostMalloc((void**)&c->h_rec, 2 * (kProsacRecCap + 1) * sizeof(uint32_t)));
        c->cap_recflags = bytes;
    }
    if (chunk > c->cap_chunk) {
        LR_HIP(hipStreamSynchronize(c->stream));
        if (dev_alloc(c->d_samples, 4 * chunk) || dev_alloc(c->d_hcounts, 2 * chunk)) return 1;
        if (c->h_samples) (void)hipHostFree(c->h_samples);
        if (c->h_hcounts) (void)hipHostFree(c->h_hcounts);
        c->h_samples = nullptr;
        c->h_hcounts = nullptr;
        LR_HIP(hipHostMalloc((void**)&c->h_samples, 4 * chunk * sizeof(uint32_t)));
        LR_HIP(hipHostMalloc((void**)&c->h_hcounts, 2 * chunk * sizeof(uint32_t)));
        c->cap_chunk = chunk;
    }
    for (auto& e : c->prosac_ev)
        if (!e) LR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return 0;
}

// Stable argsort by weight, descending (reference utils.h:36-44 uses std::stable_sort with a > comparator).  The weights
// are fourth powers (>= +0), so the order of their bit patterns is their order: three stable 11-bit counting passes over
// the complemented bits, a fifth of std::stable_sort's time on 24 000 lines.  A NaN weight (a line through the peak
// itself) has no place in that order: then the comparison sort decides, as before.
void stable_order_descending(const std::vector<float>& w, std::vector<int>& order) {
    const size_t n = w.size();
    order.resize(n);
    bool plain = true;
    for (size_t i = 0; i < n; ++i) plain = plain && w[i] >= 0.0f && !std::signbit(w[i]);  // (false for NaN and -0)
    if (!plain || n < 256) {
        for (size_t i = 0; i < n; ++i) order[i] = (int)i;
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
        return;
    }
    std::vector<uint32_t> key(n), key2(n);
    std::vector<int> idx(n), idx2(n);
    for (size_t i = 0; i < n; ++i) {
        uint32_t b;
        std::memcpy(&b, &w[i], 4);
        key[i] = ~b;
        idx[i] = (int)i;
    }
    for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass * 11;
        uint32_t cnt[2049] = {0};
        for (size_t i = 0; i < n; ++i) cnt[((key[i] >> shift) & 2047u) + 1]++;
        for (int b = 0; b < 2048; ++b) cnt[b + 1] += cnt[b];
        for (size_t i = 0; i < n; ++i) {
            const uint32_t p = cnt[(key[i] >> shift) & 2047u]++;
            key2[p] = key[i];
            idx2[p] = idx[i];
        }
        key.swap(key2);
        idx.swap(idx2);
    }
    order = idx;
}

// prosac.h:31-55
int niter_ransac(double p, double epsilon, int s, int Nmax) {
    if (Nmax == -1) Nmax = INT32_MAX;
    if (epsilon <= 0.) return 1;
    const double logarg = -std::exp(s * std::log(1. - epsilon));
    const double logval = std::log(1. + logarg);
    const double N = std::log(1. - p) / logval;
    if (logval < 0. && N < Nmax) return (int)std::ceil(N);
    return Nmax;
}

const float kChi2[20] = {INFINITY,   6.6348966f,  5.41189443f, 4.70929225f, 4.21788459f, 3.84145882f, 3.5373846f,
                         3.28302029f, 3.06490172f, 2.8743734f,  2.70554345f, 2.55422131f, 2.41732093f, 2.29250453f,
                         2.17795916f, 2.07225086f, 1.97422609f, 1.88294329f, 1.79762406f, 1.71761761f};

inline uint32_t sample_one(uint64_t seed, uint32_t round, uint32_t iter, uint32_t n) {
    const uint64_t z = splitmix64(seed ^ splitmix64(((uint64_t)round << 32) | iter));
    return (uint32_t)(((uint64_t)(uint32_t)z * n) >> 32);
}

// the growth function of PROSAC (prosac.h:150-166): pure bookkeeping, no data
struct Growth {
    int t, n, T_n_prime;
    double T_n;
    void advance(int n_star, int m) {
        t = t + 1;
        if ((t > T_n_prime) && (n < n_star)) {
            const double T_nplus1 = (T_n * (n + 1)) / (n + 1 - m);
            n = n + 1;
            T_n_prime = T_n_prime + (int)std::ceil(T_nplus1 - T_n);
            T_n = T_nplus1;
        }
    }
};

}  // namespace

// get_weights (line_pencil.cpp:47-86): vote pairs from the host's std::mt19937 (default seed, as the
// reference), accumulator and peak on the GPU, weights back on the host (positions follow `indices`).
int ctx_ht_weights(lr_context* c, const PencilModel& model, const std::vector<int>& indices, std::vector<float>& weights) {
    LR_HIP(hipSetDevice(c->device));
    const size_t n = indices.size();
    weights.assign(n, 0.f);
    if (n == 0) return 0;
    const int n_pairs = 20000, ht = 65;  // line_pencil.h:26-27
    if (ensure_prosac_buffers(c, n, (size_t)n_pairs, 1)) return 1;
    {
        std::mt19937 rng;
        std::uniform_int_distribution<int> rand_idx(0, (int)n - 1);
        for (int i = 0; i < n_pairs; ++i) {
            c->h_pairs[i] = rand_idx(rng);
            c->h_pairs[n_pairs + i] = rand_idx(rng);
        }
    }
    PencilSoA m;
    if (upload_model(c, model, indices, &m)) return 1;
    LR_HIP(hipMemcpyAsync(c->d_pairs, c->h_pairs, 2 * (size_t)n_pairs * sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
    if (launch_ht_weights(m, (uint32_t)n, c->d_pairs, c->d_pairs + n_pairs, n_pairs, ht, c->d_peak, c->d_weights, c->stream))
        return 1;
    LR_HIP(hipMemcpyAsync(c->h_weights, c->d_weights, n * sizeof(float), hipMemcpyDeviceToHost, c->stream));
    LR_HIP(hipStreamSynchronize(c->stream));
    std::copy(c->h_weights, c->h_weights + n, weights.begin());
    return 0;
}

// PROSAC_Estimator::solve (prosac.h:104-299).  The sequential loop is replayed on the host exactly as
// written; only "support of the model" (the inlier count of every sample against all lines) runs on the
// GPU, for a speculative chunk of upcoming iterations at a time.  The sample of iteration t depends on
// earlier results only through n_star, which changes when a new best hypothesis appears: the chunk is
// then cut at that iteration and the rest regenerated, so the outcome equals the sequential run.
int ctx_prosac_solve(lr_context* c, const PencilModel& model, const std::vector<int>& indices, float tol, int T_N_in,
                     uint64_t seed, uint32_t round, Vec3* h_out, ProsacTrace* trace) {
    static const bool pdebug = std::getenv("LIBRECTIFY_PROSAC_DEBUG") != nullptr;
    double t_w = now_ms(), t_events = 0, t_gen = 0, t_gpu = 0, t_flags = 0, t_len = 0;
    int n_events = 0, n_chunks = 0, n_single = 0;
    std::vector<float> weights;
    if (ctx_ht_weights(c, model, indices, weights)) return 1;
    const double t_w1 = now_ms();
    const int N = (int)indices.size();
    std::vector<int> order(N);
    stable_order_descending(weights, order);  // utils.h:36-44 (argsort, stable, by weight descending)
    const double t_s1 = now_ms();
    std::vector<int> idx(N);
    for (int i = 0; i < N; ++i) idx[i] = indices[order[i]];
    const int m = 2;
    const float eta = 0.05f, beta = 0.01f, psi = 0.02f, p_good = 0.9f, max_outlier = 0.5f;  // prosac.h:62-66
    const int T_N = T_N_in > 0 ? T_N_in : niter_ransac(p_good, max_outlier, m, -1);
    float chi2_value;
    {
        const float p2 = 2 * psi;
        chi2_value = kChi2[(int)std::floor(std::max(std::min(p2, 0.2f), 0.01f) * 100)];
    }
    auto Imin = [&](int mm, int n) {
        const double mu = n * beta;
        const double sigma = std::sqrt(n * beta * (1 - beta));
        return (int)std::ceil(mm + mu + sigma * std::sqrt(chi2_value));
    };
    int n_star = N, I_n_star = 0, I_N_best = 0, k_n_star = T_N, best_iter = -1;
    const int I_N_min = (int)((1. - max_outlier) * N);
    Growth g{0, m, 1, (double)T_N};
    for (int i = 0; i < m; i++) g.T_n *= (double)(g.n - i) / (N - i);
    Vec3 p_best{0, 0, 0};
    std::vector<uint8_t> best_inl(N, 0), isInlier(N);
    std::vector<int> pre;
    PencilSoA soa;
    if (N >= 2 && upload_model(c, model, idx, &soa)) return 1;
    // sample of iteration s.t from the growth state (prosac.h:170-190)
    auto sample_of = [&](const Growth& s, uint32_t& sa, uint32_t& sb) {
        if (s.t > s.T_n_prime) {
            sample_pair(seed, round, (uint32_t)s.t, (uint32_t)s.n, sa, sb);
        } else {
            sa = sample_one(seed, round, (uint32_t)s.t, (uint32_t)(s.n - 1));
            sb = (uint32_t)(s.n - 1);  // prosac.h:186 writes n (one past U_n); n-1 is meant
        }
    };
    // Chunks of upcoming iterations, two in flight: while the host goes through the counts of one, the GPU works on the
    // next, which was generated as if the first held no new best.  Nothing of a chunk is used without the check below
    // (iteration by iteration: does the true state still draw this sample?), so a chunk generated under a wrong guess
    // costs GPU time and never a result.
    constexpr size_t kChunkMax = 1u << 16;
    struct Chunk {
        size_t cnt = 0;
        int buf = 0;
        Growth start{0, 0, 0, 0.0}, end{0, 0, 0, 0.0};  // growth state before its first / behind its last sample
        int n_star = 0;                                  // ... and the n_star it was drawn with
        bool live = false;
    };
    if (N >= 2 && ensure_prosac_buffers(c, (size_t)N, 1, kChunkMax)) return 1;
    const size_t rf_bytes = c->cap_recflags;
    auto running = [&](const Growth& s) { return ((I_N_best < I_N_min) || s.t <= k_n_star) && s.t < T_N; };
    size_t chunk = 2048;
    auto start_chunk = [&](const Growth& from, int buf, Chunk& ch) -> int {
        const double tg0 = now_ms();
        uint32_t* hs = c->h_samples + (size_t)buf * 2 * kChunkMax;
        uint32_t* ds = c->d_samples + (size_t)buf * 2 * kChunkMax;
        Growth s = from;
        // the second sample of the pairs follows the first ones directly: the length is not known before the loop ends,
        // so they are written at the far end first (cheap: one pass over 4 bytes per iteration)
        size_t cnt = 0;
        while (cnt < chunk && running(s)) {
            s.advance(n_star, m);
            uint32_t sa, sb;
            sample_of(s, sa, sb);
            hs[cnt] = sa;
            hs[kChunkMax + cnt] = sb;
            ++cnt;
        }
        ch.cnt = cnt;
        ch.buf = buf;
        ch.start = from;
        ch.end = s;
        ch.n_star = n_star;
        ch.live = cnt > 0;
        t_gen += now_ms() - tg0;
        if (!ch.live) return 0;
        ++n_chunks;
        if (cnt < kChunkMax) std::memmove(hs + cnt, hs + kChunkMax, cnt * sizeof(uint32_t));
        uint32_t* dcnt = c->d_hcounts + (size_t)buf * kChunkMax;
        uint32_t* drec = c->d_rec + (size_t)buf * (kProsacRecCap + 1);
        uint8_t* dflags = c->d_recflags + (size_t)buf * rf_bytes;
        LR_HIP(hipMemcpyAsync(ds, hs, 2 * cnt * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
        if (launch_prosac_count(soa, (uint32_t)N, tol, model.degeneracy_tol, ds, ds + cnt, (uint32_t)cnt, dcnt, c->stream)) return 1;
        LR_HIP(hipMemcpyAsync(c->h_hcounts + (size_t)buf * kChunkMax, dcnt, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        // the chunk's new-best iterations (they follow from the counts and the best count known when it is enqueued: a
        // best found in the chunk before it can only strike some of them off) and the inlier flags of each come back
        // with the counts: one wait per chunk, not one per new best
        if (launch_prosac_records(soa, (uint32_t)N, ds, ds + cnt, dcnt, (uint32_t)cnt, (uint32_t)std::max(I_N_best, 0), tol, drec,
                                  kProsacRecCap, dflags, c->stream))
            return 1;
        LR_HIP(hipMemcpyAsync(c->h_rec + (size_t)buf * (kProsacRecCap + 1), drec, (kProsacRecCap + 1) * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        LR_HIP(hipMemcpyAsync(c->h_recflags + (size_t)buf * rf_bytes, dflags, (size_t)N * kProsacRecCap, hipMemcpyDeviceToHost, c->stream));
        LR_HIP(hipEventRecord(c->prosac_ev[buf], c->stream));
        chunk = std::min<size_t>(chunk * 4, kChunkMax);
        return 0;
    };
    Chunk cur, nxt;
    while (N >= 2 && running(g)) {
        if (!cur.live) {
            if (start_chunk(g, 0, cur)) return 1;
            if (!cur.live) break;
        }
        if (!nxt.live && start_chunk(cur.end, cur.buf ^ 1, nxt)) return 1;
        const double tg1 = now_ms();
        LR_HIP(hipEventSynchronize(c->prosac_ev[cur.buf]));
        const double tg2 = now_ms();
        t_gpu += tg2 - tg1;
        const uint32_t* hs = c->h_samples + (size_t)cur.buf * 2 * kChunkMax;
        const uint32_t* hsb = hs + cur.cnt;
        const uint32_t* hcnt = c->h_hcounts + (size_t)cur.buf * kChunkMax;
        const uint32_t* hrec = c->h_rec + (size_t)cur.buf * (kProsacRecCap + 1);
        const uint8_t* hflags = c->h_recflags + (size_t)cur.buf * rf_bytes;
        const uint32_t n_rec = std::min<uint32_t>(hrec[0], kProsacRecCap);
        uint32_t rec_pos = 0;
        // The chunk was generated under the state at its start (or an earlier one).  A new best hypothesis changes n_star
        // and k_n_star; what follows it in the chunk is still the sequential algorithm's as long as the loop would go on
        // and would draw the same sample: checked iteration by iteration, and the chunk is cut where that stops being true.
        // (Drawn from this very state with this n_star, the samples ARE the sequence: nothing to compare until a new best
        // changes n_star.)
        bool same = cur.n_star == n_star && cur.start.t == g.t && cur.start.n == g.n && cur.start.T_n_prime == g.T_n_prime &&
                    cur.start.T_n == g.T_n;
        size_t j = 0;
        for (; j < cur.cnt; ++j) {
            if (!running(g)) break;
            Growth gn = g;
            gn.advance(n_star, m);
            if (!same) {
                uint32_t ea, eb;
                sample_of(gn, ea, eb);
                if (ea != hs[j] || eb != hsb[j]) break;
            }
            g = gn;
            const uint32_t I = hcnt[j];
            if (I == 0xFFFFFFFFu) continue;  // degenerate sample
            if ((int)I > I_N_best) {
                const int ia = idx[hs[j]], ib = idx[hsb[j]];
                const Vec3 p_t = model.fit(ia, ib);
                int I_N = 0;
                const double te0 = pdebug ? now_ms() : 0.;
                while (rec_pos < n_rec && hrec[1 + rec_pos] < (uint32_t)j) ++rec_pos;
                if (rec_pos < n_rec && hrec[1 + rec_pos] == (uint32_t)j) {  // its flags came with the chunk
                    std::memcpy(isInlier.data(), hflags + (size_t)rec_pos * N, (size_t)N);
                    for (int i = 0; i < N; ++i) I_N += isInlier[i];
                } else if (N >= 4096) {  // (more new bests in the chunk than flag rows: one by one)
                    if (launch_prosac_flags(soa, (uint32_t)N, p_t.x, p_t.y, p_t.z, tol, reinterpret_cast<uint8_t*>(c->d_weights), c->stream))  // (the weights buffer is free by now)
                        return 1;
                    LR_HIP(hipMemcpyAsync(isInlier.data(), c->d_weights, (size_t)N, hipMemcpyDeviceToHost, c->stream));
                    LR_HIP(hipStreamSynchronize(c->stream));
                    for (int i = 0; i < N; ++i) I_N += isInlier[i];
                    ++n_single;
                } else {
                    for (int i = 0; i < N; ++i) {
                        isInlier[i] = model.error(p_t, idx[i]) < tol;
                        I_N += isInlier[i];
                    }
                }
                I_N_best = I_N;
                p_best = p_t;
                best_inl = isInlier;
                best_iter = g.t;
                const double te1 = pdebug ? now_ms() : 0.;
                t_flags += te1 - te0;
                int n_best = N, I_n_best = I_N;
                double epsilon_n_best = (double)I_n_best / n_best;
                // prosac.h:236-262, the search for the best termination length, as written -- but lengths that could pass
                // its two conditions (more inliers per line among the first n_test than among the first n_best, and
                // more than chance explains) are looked for 64 at a time on the prefix counts (a loop without exits,
                // which the compiler vectorises): few lengths do, and the scalar loop took a square root for most of
                // the N of them, for every new best
                // (Imin depends on the length alone, not on the lines: two roots per length once per context, not per new best)
                std::vector<int>& imin_tab = c->prosac_imin;
                for (int n = (int)imin_tab.size(); n <= N; ++n) imin_tab.push_back(n > m ? Imin(m, n) : 0);
                pre.resize((size_t)N + 1);
                pre[0] = 0;
                for (int i = 0; i < N; ++i) pre[(size_t)i + 1] = pre[(size_t)i] + isInlier[i];
                int n_test = N;
                bool stop = false;
                while (n_test > m && !stop) {
                    const int lo = std::max(m + 1, n_test - 63);
                    // (the second condition without its root, in single precision with room for every rounding -- the sum
                    // I - eps n is off by less than 4e-7 n + 0.02 for any line count the interface allows: lengths whose
                    // upper bound of (I - eps n)^2 is clearly below the variance term cannot pass; the expression as
                    // written decides in the scalar loop.  Four lengths per SSE instruction.)
                    const float e_b = (float)epsilon_n_best;
                    const float q_b = (float)(epsilon_n_best * (1. - epsilon_n_best) * 2.706 * (1. - 1e-3));
                    int any = 0;
                    for (int n = lo; n <= n_test; ++n) {
                        const float dn = (float)n, dd = ((float)pre[(size_t)n] - e_b * dn) + (4e-7f * dn + 0.02f);
                        any |= (dd > 0.f) & (dd * dd >= dn * q_b);
                    }
                    if (!any) {
                        n_test = lo - 1;
                        continue;
                    }
                    for (; n_test >= lo; n_test--) {
                        const int I_n_test = pre[(size_t)n_test];
                        if (!(I_n_test * n_best > I_n_best * n_test)) continue;
                        // I > eps n + sqrt(v): decided on (I - eps n)^2 against v where that is clear of every rounding,
                        // by the expression as written otherwise (the root is what this loop's time went into)
                        const double en = epsilon_n_best * n_test, v = n_test * epsilon_n_best * (1. - epsilon_n_best) * 2.706;
                        const double dd = (double)I_n_test - en;
                        bool second;
                        if (!(dd > 0.) || dd * dd < v * (1. - 1e-9)) second = false;
                        else if (dd * dd > v * (1. + 1e-9)) second = true;
                        else second = I_n_test > en + std::sqrt(v);
                        if (second) {
                            if (I_n_test < imin_tab[(size_t)n_test]) {
                                stop = true;
                                break;
                            }
                            n_best = n_test;
                            I_n_best = I_n_test;
                            epsilon_n_best = (double)I_n_best / n_best;
                        }
                    }
                }
                if (pdebug) t_len += now_ms() - te1;
                if (I_n_best * n_star > I_n_star * n_best) {
                    same = same && n_best == n_star;
                    n_star = n_best;
                    I_n_star = I_n_best;
                    k_n_star = niter_ransac(1. - eta, 1. - I_n_star / (double)n_star, m, T_N);
                }
                ++n_events;
            }
        }
        t_events += now_ms() - tg2;
        if (j == cur.cnt && nxt.live) {
            cur = nxt;  // its first iteration is the one after this chunk's last: still in step
            nxt.live = false;
        } else {
            // cut (or over): what is in flight continues a sequence that was not drawn; its buffers are free once it is done
            if (nxt.live) LR_HIP(hipEventSynchronize(c->prosac_ev[nxt.buf]));
            cur.live = nxt.live = false;
        }
    }
    if (cur.live || nxt.live) LR_HIP(hipStreamSynchronize(c->stream));
    if (pdebug)
        std::fprintf(stderr, "prosac round %u: N %d, weights %.2f ms, sort %.2f, chunks %d (generate %.2f, gpu+sync %.2f, scan+events %.2f of which flags %.2f, lengths %.2f; %d events, %d with a wait of their own), total %.2f ms\n",
                     round, N, t_w1 - t_w, t_s1 - t_w1, n_chunks, t_gen, t_gpu, t_events, t_flags, t_len, n_events, n_single, now_ms() - t_w);
    if (trace) {
        trace->iterations = g.t;
        trace->n_star = n_star;
        trace->best_iter = best_iter;
        trace->I_N_best = I_N_best;
    }
    std::vector<int> inl;
    for (int i = 0; i < N; ++i)
        if (best_inl[i]) inl.push_back(idx[i]);
    *h_out = model.fit_optimal(inl);
    return 0;
}

int ctx_estimate_line_pencils_prosac(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                                     float garbage_deg, int T_N, uint64_t seed) {
    if (lines.empty()) return 0;
    const Normalisation nrm = bbox_normalisation(lines);
    const PencilModel model(normalise(lines, nrm));
    const float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    const int N = model.size();
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int remaining = N, k = 0;
    while (remaining >= 2 && k < max_models) {
        std::vector<int> obs;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0) obs.push_back(i);
        Vec3 h;
        if (ctx_prosac_solve(c, model, obs, tol, T_N, seed, (uint32_t)k, &h, nullptr)) return 1;
        int n_in = 0, n_gb = 0;
        for (int i : obs) {
            const float e = model.error(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
            }
        }
        remaining -= n_in + n_gb;
        ++k;
    }
    for (int i = 0; i < N; ++i) lines[i].group_id = garbage_flag[i] == 1 ? -1 : inlier_flag[i];
    return 0;
}

// DirectEstimator (estimator.h:82-96; compiled by the reference, never instantiated): the lines whose Hough weight
// (GPU: ctx_ht_weights) exceeds 0.95 decide the refit; an empty set means every line (line_pencil.cpp:114-117).
int ctx_direct_solve(lr_context* c, const PencilModel& model, const std::vector<int>& indices, Vec3* h) {
    std::vector<float> weights;
    if (ctx_ht_weights(c, model, indices, weights)) return 1;
    std::vector<int> inl;
    for (size_t j = 0; j < indices.size(); ++j)
        if (weights[j] > 0.95f) inl.push_back(indices[j]);
    *h = model.fit_optimal(inl);
    return 0;
}

int ctx_estimate_line_pencils_direct(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                                     float garbage_deg) {
    if (lines.empty()) return 0;
    const Normalisation nrm = bbox_normalisation(lines);
    const PencilModel model(normalise(lines, nrm));
    const float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    const int N = model.size();
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int remaining = N, k = 0;
    while (remaining >= 2 && k < max_models) {
        std::vector<int> obs;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0) obs.push_back(i);
        Vec3 h;
        if (ctx_direct_solve(c, model, obs, &h)) return 1;
        int n_in = 0, n_gb = 0;
        for (int i : obs) {
            const float e = model.error(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
            }
        }
        remaining -= n_in + n_gb;
        ++k;
    }
    for (int i = 0; i < N; ++i) lines[i].group_id = garbage_flag[i] == 1 ? -1 : inlier_flag[i];
    return 0;
}

// Diamond-space accumulator (opt-in; cht.h:13-24): de-normalised vanishing point of the strongest pencil.
int ctx_cht_vanishing_point(lr_context* c, const std::vector<LineSegment>& lines, int d, Vec3* vp,
                            std::vector<uint64_t>* acc_out) {
    LR_HIP(hipSetDevice(c->device));
    const Normalisation nrm = bbox_normalisation(lines);
    const PencilModel model(normalise(lines, nrm));
    std::vector<int> all(model.size());
    for (int i = 0; i < model.size(); ++i) all[i] = i;
    PencilSoA soa;
    if (upload_model(c, model, all, &soa)) return 1;
    const size_t cells = (size_t)d * d;
    if (cells > c->cap_cht) {  // accumulator kept in the context (no allocation on the path of a call)
        LR_HIP(hipStreamSynchronize(c->stream));
        if (dev_alloc(c->d_cht_acc, cells)) return 1;
        c->cap_cht = cells;
    }
    if (launch_cht_accumulate(soa, (uint32_t)model.size(), d, c->d_cht_acc, c->stream)) return 1;
    std::vector<uint64_t> acc(cells);
    LR_HIP(hipMemcpyAsync(acc.data(), c->d_cht_acc, acc.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
    LR_HIP(hipStreamSynchronize(c->stream));
    size_t best = 0;
    for (size_t i = 1; i < acc.size(); ++i)
        if (acc[i] > acc[best]) best = i;
    const int iy = (int)(best / d), ix = (int)(best % d);
    const float u = (float)ix / (float)(d - 1) * 2.f - 1.f, v = (float)iy / (float)(d - 1) * 2.f - 1.f;
    const float su = u >= 0.f ? 1.f : -1.f, sv = v >= 0.f ? 1.f : -1.f;
    Vec3 p{v, su * u + sv * v - 1.f, u};
    if (std::fabs(p.z) < kEps) {
        p.z = 0.f;
    } else {
        p = {p.x / p.z, p.y / p.z, 1.f};
        p.x = nrm.scale * p.x + nrm.center.x;
        p.y = nrm.scale * p.y + nrm.center.y;
    }
    *vp = p;
    if (acc_out) acc_out->swap(acc);
    return 0;
}

// The diamond-space accumulator as an ESTIMATOR of the path (opt-in, lr_set_estimator(3, d); cht.h:13-24 describes
// accumulate -> argmax -> de-normalise, "the weights can be negative (so lines can be removed!)"): the peeling loop of
// estimate_multiple_structures (estimator.h:99-145) around a solve() that reads
//     hypothesis = point of the accumulator's strongest cell (first maximum in row-major order)
//     inliers    = remaining lines whose inclination error against it is below tol   (as estimator.h:74)
//     model      = fit_optimal(inliers)                                              (as estimator.h:75-76)
// The votes of every line go into the accumulator once; after a round the lines it has grouped or discarded are taken
// back out of it with negative votes (exact: the votes are integers), instead of accumulating the rest again -- the
// oracle re-accumulates, so the two check each other.  Accumulation and argmax run on the GPU, one 12-byte peak comes
// back per round; the O(n) verdicts stay on the host like PROSAC's and Direct's.  Parity unpinned: the reference's
// cht.cpp does not compile.
int ctx_estimate_line_pencils_cht(lr_context* c, std::vector<LineSegment>& lines, int max_models, float inlier_deg,
                                  float garbage_deg, int d, ChtTrace* trace) {
    if (lines.empty()) return 0;
    LR_HIP(hipSetDevice(c->device));
    if (d <= 0) d = 128;
    const Normalisation nrm = bbox_normalisation(lines);
    const PencilModel model(normalise(lines, nrm));
    const float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    const int N = model.size();
    std::vector<int> all(N);
    for (int i = 0; i < N; ++i) all[i] = i;
    PencilSoA soa;
    if (upload_model(c, model, all, &soa)) return 1;
    const size_t cells = (size_t)d * d;
    if (cells > c->cap_cht) {
        LR_HIP(hipStreamSynchronize(c->stream));
        if (dev_alloc(c->d_cht_acc, cells)) return 1;
        c->cap_cht = cells;
    }
    if ((size_t)N > c->cap_cht_idx) {
        LR_HIP(hipStreamSynchronize(c->stream));
        const size_t cl = std::max<size_t>((size_t)N, 4096);
        if (dev_alloc(c->d_cht_idx, cl)) return 1;
        if (c->h_cht_idx) (void)hipHostFree(c->h_cht_idx);
        c->h_cht_idx = nullptr;
        LR_HIP(hipHostMalloc((void**)&c->h_cht_idx, cl * sizeof(uint32_t)));
        c->cap_cht_idx = cl;
    }
    if (!c->d_cht_peak) {
        if (dev_alloc(c->d_cht_peak, 8)) return 1;
        LR_HIP(hipHostMalloc((void**)&c->h_cht_peak, 8 * sizeof(uint32_t)));
    }
    unsigned long long* d_votes = reinterpret_cast<unsigned long long*>(c->d_cht_peak + 4);
    LR_HIP(hipMemsetAsync(c->d_cht_peak, 0, 8 * sizeof(uint32_t), c->stream));
    LR_HIP(hipMemsetAsync(c->d_cht_acc, 0, cells * sizeof(unsigned long long), c->stream));
    if (launch_cht_votes(soa, nullptr, (uint32_t)N, d, c->d_cht_acc, false, d_votes, c->stream)) return 1;
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int remaining = N, k = 0;
    while (remaining >= 2 && k < max_models) {  // estimator.h:115
        if (launch_cht_peak(c->d_cht_acc, d, c->d_cht_peak, c->stream)) return 1;
        LR_HIP(hipMemcpyAsync(c->h_cht_peak, c->d_cht_peak, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        LR_HIP(hipStreamSynchronize(c->stream));
        const uint32_t cell = c->h_cht_peak[0];
        const int iy = (int)(cell / (uint32_t)d), ix = (int)(cell % (uint32_t)d);
        const float u = (float)ix / (float)(d - 1) * 2.f - 1.f, v = (float)iy / (float)(d - 1) * 2.f - 1.f;
        const Vec3 p{v, (u >= 0.f ? 1.f : -1.f) * u + (v >= 0.f ? 1.f : -1.f) * v - 1.f, u};
        std::vector<int> inl;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0 && model.error(p, i) < tol) inl.push_back(i);
        const Vec3 h = model.fit_optimal(inl);
        if (trace) {
            trace->models.push_back(h);
            trace->peak_cell.push_back(cell);
        }
        uint32_t n_out = 0;
        int n_in = 0, n_gb = 0;
        for (int i = 0; i < N; ++i) {
            if (inlier_flag[i] >= 0 || garbage_flag[i] != 0) continue;
            const float e = model.error(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
                c->h_cht_idx[n_out++] = (uint32_t)i;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
                c->h_cht_idx[n_out++] = (uint32_t)i;
            }
        }
        remaining -= n_in + n_gb;
        ++k;
        if (remaining >= 2 && k < max_models && n_out > 0) {  // the next round votes without them
            LR_HIP(hipMemcpyAsync(c->d_cht_idx, c->h_cht_idx, (size_t)n_out * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream));
            if (launch_cht_votes(soa, c->d_cht_idx, n_out, d, c->d_cht_acc, true, d_votes, c->stream)) return 1;
        }
    }
    if (trace) {
        LR_HIP(hipMemcpyAsync(c->h_cht_peak, c->d_cht_peak, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        LR_HIP(hipStreamSynchronize(c->stream));
        trace->votes = ((uint64_t)c->h_cht_peak[5] << 32) | c->h_cht_peak[4];
    }
    for (int i = 0; i < N; ++i) lines[i].group_id = garbage_flag[i] == 1 ? -1 : inlier_flag[i];
    return 0;
}

// postprocess_lines_segments (line_detector.cpp:332-444): pair test on the GPU for large n, graph walk and
// merges on the host.
int ctx_refine(lr_context* c, std::vector<LineSegment>& lines) {
    const size_t n = lines.size();
    if (n < 2048) {
        lines = refine_lines(lines);
        return 0;
    }
    LR_HIP(hipSetDevice(c->device));
    std::vector<float> table;
    refine_segment_table(lines, table);
    // segment table and edge list live in the context and grow on demand (no allocation on the path of a call)
    if (table.size() > c->cap_refine_table) {
        LR_HIP(hipStreamSynchronize(c->stream));
        if (dev_alloc(c->d_refine_table, table.size())) return 1;
        c->cap_refine_table = table.size();
    }
    size_t cap = std::max<size_t>(16 * n, c->cap_refine_edges);
    std::vector<std::pair<uint32_t, uint32_t>> edges;
    static_assert(sizeof(std::pair<uint32_t, uint32_t>) == sizeof(uint2), "edge layout");
    LR_HIP(hipMemcpyAsync(c->d_refine_table, table.data(), table.size() * sizeof(float), hipMemcpyHostToDevice, c->stream));
    for (;;) {
        if (cap > c->cap_refine_edges) {
            LR_HIP(hipStreamSynchronize(c->stream));
            uint2* e = static_cast<uint2*>(c->d_refine_edges);
            if (dev_alloc(e, cap)) return 1;
            c->d_refine_edges = e;
            c->cap_refine_edges = cap;
        }
        LR_HIP(hipMemsetAsync(c->d_counts + 8, 0, sizeof(uint32_t), c->stream));
        if (launch_refine_pairs(c->d_refine_table, (uint32_t)n, c->d_refine_edges, c->d_counts + 8,
                                (uint32_t)std::min<size_t>(c->cap_refine_edges, 0xFFFFFFFFu), c->stream))
            return 1;
        LR_HIP(hipMemcpyAsync(c->h_counts + 8, c->d_counts + 8, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        LR_HIP(hipStreamSynchronize(c->stream));
        const size_t ne = c->h_counts[8];
        if (ne <= c->cap_refine_edges) {
            edges.resize(ne);
            if (ne) {
                LR_HIP(hipMemcpyAsync(edges.data(), c->d_refine_edges, ne * sizeof(uint2), hipMemcpyDeviceToHost, c->stream));
                LR_HIP(hipStreamSynchronize(c->stream));
            }
            break;
        }
        cap = ne;  // the kernel counted every edge: exactly enough next time
    }
    lines = refine_lines_from_edges(lines, edges);
    return 0;
}

// find_line_segment_groups (interface.cpp:35-80) on a device-resident image.
//
// Default path (RANSAC, refine off): every stage of the frame -- filter, seeds, flood rounds, line fit, filter_lines,
// the four peeling rounds -- is enqueued without a single host round trip, then the counts and the grouped lines come
// back in one copy and the host waits ONCE.  Two things can make a frame take a second lap, both rare and both
// detected from that copy: more seeds than the seed sort was sized for (the frame is repeated with room), and a flood
// that needs more rounds than were enqueued blindly (the rounds are completed, the stages after the flood repeated).
// With refine or PROSAC the raw segments go to the host after the fit, as before.
static int run_frame(lr_context* c, const float* d_image, int w, int h, int stride, float min_length, bool refine,
                     std::vector<LineSegment>& out, bool filter_enqueued = false) {
    out.clear();
    const double t_begin = now_ms();
    const bool fused = !refine && c->estimator == 0;
    const int n_iter = c->ransac_iters;
    auto groups_after_fit = [&]() -> int {
        const uint32_t lc = line_cap_for(c);
        if (ensure_group_capacity(c, lc, (size_t)std::max(n_iter, 1))) return 1;
        if (ensure_result_block(c, std::max<size_t>(c->res_lines_cap, 4096))) return 1;
        if (c->timing_on) LR_HIP(hipEventRecord(c->ev[5], c->stream));
        const PencilTable all = table_of(c, 0), round0 = table_of(c, 1);
        if (launch_filter_lines(c->d_lines, c->d_counts + kCntComp, lc, min_length, c->d_flines, c->d_gctl, c->d_gnorm, &all, &round0,
                                c->stream))
            return 1;
        static_assert(kMaxModels >= 1 && kGcWords == 8 && kResHeaderBytes == 256, "the last peeling round carries the result block (peel_kernel)");
        if (enqueue_groups(c, lc, kMaxModels, kInlierDeg, kGarbageDeg, n_iter, c->ransac_seed, true, true)) return 1;
        if (c->timing_on) LR_HIP(hipEventRecord(c->ev[6], c->stream));
        return 0;
    };
    c->frame_laps = 0;
    for (int attempt = 0;; ++attempt) {
        c->frame_laps += 1;
        // (a frame that came from a host buffer has had its filter launched band by band as its rows arrived:
        // ctx_find_groups_host; a second lap takes the whole frame from the device slot)
        if (!(filter_enqueued && attempt == 0) && enqueue_filter(c, d_image, w, h, stride)) return 1;
        if (enqueue_seeds(c)) return 1;
        if (enqueue_flood(c)) return 1;
        if (enqueue_fit(c)) return 1;
        if (fused) {
            if (groups_after_fit()) return 1;
        } else {
            LR_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        }
        const double t_enq = now_ms();
        const double t_pre = now_ms();
        if (c->sleep_in_wait) {  // a batch lane: leave the core to the threads that stage frames
            LR_HIP(hipEventRecord(c->ev_wait, c->stream));
            LR_HIP(hipEventSynchronize(c->ev_wait));
        } else {
            LR_HIP(hipStreamSynchronize(c->stream));
        }
        c->host_ms[0] = t_enq - t_begin;      // enqueue of the frame's kernels
        c->host_ms[1] = t_pre - t_enq;        // staging + upload of the lane's next frame
        c->host_ms[2] = now_ms() - t_pre;     // wait for the GPU
        const uint32_t* cnt = fused ? reinterpret_cast<const uint32_t*>(c->h_res) : c->h_counts;
        c->n_seeds = cnt[kCntSeeds];
        if (c->n_seeds <= c->seed_cap) break;
        if (attempt > 0) {
            set_error("seed count exceeds the sort capacity twice");
            return 1;
        }
        adapt_seed_cap(c, c->n_seeds);
    }
    bool extra = false;
    if (finish_flood(c, &extra)) return 1;
    if (extra) {  // the label image changed after the fit ran: the stages after the flood again
        c->frame_laps += 1;
        if (enqueue_fit(c)) return 1;
        if (fused) {
            if (groups_after_fit()) return 1;
        } else {
            LR_HIP(hipMemcpyAsync(c->h_counts, c->d_counts, 8 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        }
        LR_HIP(hipStreamSynchronize(c->stream));
    }
    const uint32_t* cnt = fused ? reinterpret_cast<const uint32_t*>(c->h_res) : c->h_counts;
    c->n_comp = cnt[kCntComp];
    c->n_px = cnt[kCntPx];
    adapt_seed_cap(c, c->n_seeds);
    c->stage_valid[0] = c->stage_valid[1] = false;
    c->stage_valid[2] = c->stage_valid[3] = true;
    if (fused) {
        const uint32_t* gctl = reinterpret_cast<const uint32_t*>(c->h_res + 32);
        const size_t n = gctl[kGcLines];
        out.resize(n);
        const size_t have = std::min<size_t>(n, c->res_lines_cap);
        if (have) std::memcpy(out.data(), c->h_res + kResHeaderBytes, have * sizeof(LineSegment));
        if (n > have) {  // more lines than the result block holds: fetch the rest, and size the block for the next frame
            LR_HIP(hipMemcpyAsync(out.data() + have, c->d_flines + have, (n - have) * sizeof(LineSegment),
                                  hipMemcpyDeviceToHost, c->stream));
            LR_HIP(hipStreamSynchronize(c->stream));
            if (ensure_result_block(c, round_up((uint32_t)(n + n / 2), 1024))) return 1;
        }
        record_stage_times(c, true);
#ifdef LR_PEEL_TIMING
        {
            const float* tm = reinterpret_cast<const float*>(c->h_res + 64) + 16;
            for (int r = 0; r < 4; ++r)
                std::fprintf(stderr, "peel round %d: best %.1f us, inliers %.1f, sums %.1f, jacobi %.1f, verdict %.1f\n", r,
                             tm[r * 5], tm[r * 5 + 1], tm[r * 5 + 2], tm[r * 5 + 3], tm[r * 5 + 4]);
        }
#endif
        return 0;
    }
    // ---- refine and / or PROSAC: raw segments to the host
    std::vector<LineSegment> raw(c->n_comp);
    if (c->n_comp) {
        LR_HIP(hipMemcpyAsync(raw.data(), c->d_lines, (size_t)c->n_comp * sizeof(LineSegment), hipMemcpyDeviceToHost,
                              c->stream));
        LR_HIP(hipStreamSynchronize(c->stream));
    }
    record_stage_times(c, false);
    if (raw.size() < 2) return 0;  // interface.cpp:50-54
    if (refine && ctx_refine(c, raw)) return 1;
    std::vector<LineSegment> filtered = filter_lines(raw, min_length);
    if (filtered.empty()) return 0;
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[5], c->stream));
    if (c->estimator == 1) {
        if (ctx_estimate_line_pencils_prosac(c, filtered, kMaxModels, kInlierDeg, kGarbageDeg, c->prosac_T_N,
                                             c->ransac_seed))
            return 1;
    } else if (c->estimator == 2) {
        if (ctx_estimate_line_pencils_direct(c, filtered, kMaxModels, kInlierDeg, kGarbageDeg)) return 1;
    } else if (c->estimator == 3) {
        if (ctx_estimate_line_pencils_cht(c, filtered, kMaxModels, kInlierDeg, kGarbageDeg, c->cht_d, nullptr)) return 1;
    } else if (ctx_estimate_line_pencils(c, filtered, kMaxModels, kInlierDeg, kGarbageDeg, n_iter, c->ransac_seed)) {
        return 1;
    }
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[6], c->stream));
    LR_HIP(hipStreamSynchronize(c->stream));
    record_stage_times(c, true);
    out.swap(filtered);
    return 0;
}

int ctx_find_groups_device(lr_context* c, const float* d_image, int w, int h, int stride, float min_length, bool refine,
                           std::vector<LineSegment>& out) {
    const int rc = run_frame(c, d_image, w, h, stride, min_length, refine, out);
    return rc;
}

// find_line_segment_groups on a HOST buffer (the reference's only kind of input: interface.cpp:43-48, image.cpp:11-19),
// one frame.  The upload, the filter and the host's enqueueing of the rest of the frame overlap:
//  - the frame goes up in 4 MB row bands (pageable memory through the page-locked staging buffer, filled by the
//    context's staging threads; page-locked memory straight from where it lies), an event after every band;
//  - the filter is launched band by band: a band row of the filter reads the image rows 30 by - 4 .. 30 by + 33, so the
//    band rows whose last image row lies in upload band k are launched as soon as that band's transfer is enqueued,
//    behind a wait for its event -- when the last transfer ends, all but the last ninth of the filter has run;
//  - the calling thread only drives (waits for "band k enqueued", launches its filter rows) and then enqueues the rest
//    of the frame while the last transfers are still on the link.
// What cannot overlap: everything after the filter needs the frame's largest magnitude, i.e. the whole frame.
int ctx_find_groups_host(lr_context* c, const float* buffer, int w, int h, int stride, float min_length, bool refine,
                         int num_threads, std::vector<LineSegment>& out) {
    const double t_call = now_ms();
    LR_HIP(hipSetDevice(c->device));
    if (w < 5 || h < 5 || buffer == nullptr) {
        set_error("image smaller than the 5x5 filter");
        return 1;
    }
    if ((stride < 0 ? -stride : stride) < w) {
        set_error("upload: |stride| smaller than the width");
        return 1;
    }
    if (ensure_copy_stream(c)) return 1;
    hipStream_t up = c->copy_stream;
    const size_t npix = (size_t)w * h;
    const int slot = 0;
    if (c->cap_slot[slot] < npix) {
        LR_HIP(hipStreamSynchronize(c->stream));
        LR_HIP(hipStreamSynchronize(up));
        if (dev_alloc(c->d_img_slot[slot], npix)) return 1;
        c->cap_slot[slot] = npix;
    }
    float* stage = nullptr;
    if (!is_page_locked(buffer)) {
        if (c->cap_stage[slot] < npix) {
            LR_HIP(hipStreamSynchronize(up));
            if (c->h_stage[slot]) (void)hipHostFree(c->h_stage[slot]);
            c->h_stage[slot] = nullptr;
            c->cap_stage[slot] = 0;
            LR_HIP(hipHostMalloc((void**)&c->h_stage[slot], npix * sizeof(float)));
            c->cap_stage[slot] = npix;
        }
        LR_HIP(hipEventSynchronize(c->ev_up[slot]));  // (the transfer that last read the staging buffer: long finished)
        stage = c->h_stage[slot];
    }
    if (prepare_frame(c, w, h)) return 1;
    // Whatever way this call ends, nothing of it may still be on the link or the GPU when it returns with an error: the
    // next call would fill the staging buffer under a transfer that still reads it (ev_up is only recorded on success).
    struct DrainOnError {
        lr_context* c;
        hipStream_t up;
        bool ok = false;
        ~DrainOnError() {
            if (ok) return;
            (void)hipStreamSynchronize(up);
            (void)hipStreamSynchronize(c->stream);
            (void)hipGetLastError();
        }
    } drain{c, up};
    const float* src = buffer;
    int sstride = stride;
    if (sstride < 0) {  // image.cpp:14-18: the same rows, addressed from the other end (no flip)
        src = buffer + (std::ptrdiff_t)(h - 1) * sstride;
        sstride = -sstride;
    }
    const size_t row_bytes = (size_t)w * sizeof(float);
    // Upload bands of 4 MB, the filter behind every band.  Smaller bands would start the link earlier (with eight staging
    // threads the first 4 MB bands are all ready at the same moment, 0.4 ms in), but every band costs about 18 us of its
    // own -- transfer submission, event, cross-stream wait -- on the stream that carries the frame: measured on 4K frames
    // 3.01 ms per call with 4 MB bands, 3.49 with 1 MB, 4.07 with 512 KB (profiles/r03_single_call_sweep.txt).
    static const size_t band_bytes = std::getenv("LIBRECTIFY_UPLOAD_BAND_KB") ? (size_t)std::max(64, std::atoi(std::getenv("LIBRECTIFY_UPLOAD_BAND_KB"))) << 10 : (size_t)4 << 20;
    static const int filter_every = std::getenv("LIBRECTIFY_FILTER_EVERY") ? std::max(1, std::atoi(std::getenv("LIBRECTIFY_FILTER_EVERY"))) : 1;
    // (a page-locked source goes up in ONE transfer unless the knob says otherwise: eight bands of 4 MB with an event each cost
    // the call more than the filter gains by starting under the transfer -- 2.07 -> 1.94 ms per 4K frame, round 5)
    static const bool band_env = std::getenv("LIBRECTIFY_UPLOAD_BAND_KB") != nullptr;
    int rpb = (int)std::max<size_t>(1, ((stage == nullptr && !band_env) ? (size_t)h * row_bytes : band_bytes) / row_bytes);
    if ((h + rpb - 1) / rpb > StagingCrew::kMaxBands) rpb = (h + StagingCrew::kMaxBands - 1) / StagingCrew::kMaxBands;  // (as StagingCrew::begin)
    const int n_bands = (h + rpb - 1) / rpb;
    while ((int)c->band_ev.size() < n_bands) {
        hipEvent_t e = nullptr;
        LR_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        c->band_ev.push_back(e);
    }
    float* dst = c->d_img_slot[slot];
    const int fb = filter_band_rows(), band_rows = (h + fb - 1) / fb;
    int by_next = 0;
    // filter rows that upload band k completes, behind that band's event
    auto filter_after_band = [&](int k) -> int {
        // the compute stream waits for EVERY band's own event: the bands are enqueued by several threads in no particular
        // order, so a later band's event says nothing about an earlier band
        LR_HIP(hipStreamWaitEvent(c->stream, c->band_ev[(size_t)k], 0));
        if ((k + 1) % filter_every != 0 && k != n_bands - 1) return 0;
        const int last_row = std::min(h, (k + 1) * rpb) - 1;  // last image row on the device once bands 0..k are
        int by_end = by_next;
        while (by_end < band_rows && std::min(h - 1, filter_band_last_row(by_end)) <= last_row) ++by_end;
        if (k == n_bands - 1) by_end = band_rows;
        if (launch_filter_rows(dst, w, h, w, c->fconsts, c->dx, c->dy, c->dmask, c->cand, c->cand_count, c->tile_max, by_next,
                               by_end, c->stream))
            return 1;
        by_next = by_end;
        return 0;
    };
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[0], c->stream));
    const int T = stage ? staging_threads(num_threads, npix * sizeof(float)) : 1;  // (the threads share every band: StagingCrew::work)
    if (T <= 1) {
        // the calling thread alone (the reference's serial mode, or a page-locked source that needs no staging)
        for (int k = 0; k < n_bands; ++k) {
            const int r0 = k * rpb, r1 = std::min(h, r0 + rpb);
            if (stage) {
                if (sstride == w) std::memcpy(stage + (size_t)r0 * w, src + (size_t)r0 * sstride, (size_t)(r1 - r0) * row_bytes);
                else
                    for (int r = r0; r < r1; ++r) std::memcpy(stage + (size_t)r * w, src + (size_t)r * sstride, row_bytes);
                LR_HIP(hipMemcpyAsync(dst + (size_t)r0 * w, stage + (size_t)r0 * w, (size_t)(r1 - r0) * row_bytes,
                                      hipMemcpyHostToDevice, up));
            } else if (sstride == w) {
                LR_HIP(hipMemcpyAsync(dst + (size_t)r0 * w, src + (size_t)r0 * w, (size_t)(r1 - r0) * row_bytes,
                                      hipMemcpyHostToDevice, up));
            } else {
                LR_HIP(hipMemcpy2DAsync(dst + (size_t)r0 * w, row_bytes, src + (size_t)r0 * sstride, (size_t)sstride * sizeof(float),
                                        row_bytes, (size_t)(r1 - r0), hipMemcpyHostToDevice, up));
            }
            LR_HIP(hipEventRecord(c->band_ev[(size_t)k], up));
            if (filter_after_band(k)) return 1;
        }
    } else {
        // the context's staging threads fill and send the bands; this thread follows them with the filter
        if (!c->crew || c->crew_helpers != T) {
            delete static_cast<StagingCrew*>(c->crew);
            StagingCrew* cr = new StagingCrew();
            cr->start(c, T);
            c->crew = cr;
            c->crew_helpers = T;
        }
        StagingCrew* cr = static_cast<StagingCrew*>(c->crew);
        std::vector<std::atomic<int>> ready((size_t)n_bands);
        for (auto& a : ready) a.store(0, std::memory_order_relaxed);
        const uint32_t g = cr->begin(dst, stage, src, w, h, sstride, up, c->band_ev.data(), ready.data(), band_bytes);
        int rc = 0;
        for (int k = 0; k < n_bands && rc == 0; ++k) {
            int spins = 0, r;
            while ((r = ready[(size_t)k].load(std::memory_order_acquire)) == 0) {
                // No helper alive (none got a device), or the helpers are starved of cores: this thread stages pieces itself
                // instead of waiting for ever for somebody else to.
                if ((cr->live.load(std::memory_order_acquire) == 0 || spins > 4096) && cr->work_one(g)) continue;
                if (++spins < 2048) std::this_thread::yield();
                else std::this_thread::sleep_for(std::chrono::microseconds(5));
            }
            if (r < 0 || filter_after_band(k)) rc = 1;
        }
        if (rc) cr->work(g);  // (the bands still unclaimed must be accounted for before finish() can return)
        if (cr->finish() || rc) {  // (every band accounted for before `ready` goes out of scope)
            if (get_error().empty()) set_error("upload: staging copy failed");
            return 1;
        }
    }
    if (c->timing_on) LR_HIP(hipEventRecord(c->ev[1], c->stream));
    LR_HIP(hipEventRecord(c->ev_up[slot], up));
    static const bool call_debug = std::getenv("LIBRECTIFY_CALL_DEBUG") != nullptr;
    const double t_up = now_ms();
    const int rc = run_frame(c, dst, w, h, w, min_length, refine, out, true);
    drain.ok = rc == 0;
    if (call_debug)
        std::fprintf(stderr, "host frame: bands staged, sent and filter launched in %.3f ms; rest of the frame enqueued in %.3f ms; waited %.3f ms; "
                     "results out in %.3f ms; whole call %.3f ms\n", t_up - t_call, c->host_ms[0], c->host_ms[2],
                     now_ms() - t_up - c->host_ms[0] - c->host_ms[2], now_ms() - t_call);
    return rc;
}

// Batch of independent frames: the stages of one frame are latency-bound (flood rounds,
// host round trips), so several frames are kept in flight, one host thread + context + HIP stream each ("lanes");
// frames are handed out dynamically (they differ in cost, and with a fixed assignment the batch ends on one lane).
//
// Host-resident frames (h_frames != nullptr) are uploaded by ONE uploader thread, in frame order, on the caller's
// copy stream, into a pool of lanes + 6 device frames: frame i goes to whichever slot is free (frames finish out of
// order: a heavy frame runs as long as three light ones, and slot i mod R would make the link wait for it), so the link
// works up to six frames ahead of the lanes instead of starting a lane's next transfer only when the lane starts a
// frame.  (With one slot ahead per lane -- the first design -- a quarter to a third of the frames of a 4K
// batch still waited for their own upload, although the link was busy only two thirds of the time: 7.5 Gpix/s where
// the same batch without the transfers ran at 9.0; LIBRECTIFY_LANE_DEBUG prints each frame's lead.)  A lane makes its
// stream wait for the frame's transfer and never touches host memory itself.
static int ensure_upload_ring(lr_context* c, int R, size_t npix, bool staging) {
    if (ensure_copy_stream(c)) return 1;
    if ((int)c->ring_img.size() < R || c->ring_cap_pix < npix) {
        LR_HIP(hipStreamSynchronize(c->copy_stream));
        for (float* p : c->ring_img)
            if (p) (void)hipFree(p);
        c->ring_img.assign((size_t)std::max<int>(R, (int)c->ring_img.size()), nullptr);
        c->ring_cap_pix = 0;
        const size_t cap = std::max(npix, c->ring_cap_pix);
        // (LIBRECTIFY_RING_UNCACHED=1, experiment of round 5: the frames' device buffers as uncached / fine-grained memory -- do the
        // lanes' kernels lose less beside the transfers when the incoming frames bypass the caches?  profiles/r05_dma_interference.txt)
        static const int ring_flags = std::getenv("LIBRECTIFY_RING_UNCACHED") ? std::atoi(std::getenv("LIBRECTIFY_RING_UNCACHED")) : 0;
        for (float*& p : c->ring_img) {
            if (ring_flags == 1) LR_HIP(hipExtMallocWithFlags((void**)&p, cap * sizeof(float), hipDeviceMallocUncached));
            else if (ring_flags == 2) LR_HIP(hipExtMallocWithFlags((void**)&p, cap * sizeof(float), hipDeviceMallocFinegrained));
            else LR_HIP(hipMalloc((void**)&p, cap * sizeof(float)));
        }
        c->ring_cap_pix = cap;
    }
    static const bool lane_debug = std::getenv("LIBRECTIFY_LANE_DEBUG") != nullptr || std::getenv("LIBRECTIFY_BATCH_STATS") != nullptr;  // (they time the uploads)
    while ((int)c->ring_ev.size() < (int)c->ring_img.size()) {
        hipEvent_t e = nullptr;
        LR_HIP(hipEventCreateWithFlags(&e, lane_debug ? hipEventDefault : hipEventDisableTiming));
        c->ring_ev.push_back(e);
    }
    if (staging && ((int)c->ring_stage.size() < (int)c->ring_img.size() || c->ring_stage_cap_pix < npix)) {
        LR_HIP(hipStreamSynchronize(c->copy_stream));
        for (float* p : c->ring_stage)
            if (p) (void)hipHostFree(p);
        c->ring_stage.assign(c->ring_img.size(), nullptr);
        c->ring_stage_cap_pix = 0;
        for (float*& p : c->ring_stage) LR_HIP(hipHostMalloc((void**)&p, npix * sizeof(float)));
        c->ring_stage_cap_pix = npix;
    }
    return 0;
}

static int find_groups_batch(lr_context* c, const float* d_images, size_t image_stride, const float* const* h_frames,
                             int batch, int w, int h, int stride, float min_length, bool refine, int num_threads,
                             LineSegment* out, int capacity, int* n_lines, const RectificationConfig* cfg,
                             ImageTransform* transforms) {
    if (batch <= 0) return 0;
    const int S = std::max(1, std::min(c->batch_streams, batch));
    while ((int)c->workers.size() < S - 1) {
        lr_context* wc = nullptr;
        if (ctx_create(c->device, &wc)) return 1;
        c->workers.push_back(wc);
    }
    std::vector<lr_context*> lanes;
    lanes.push_back(c);
    for (int i = 0; i < S - 1; ++i) lanes.push_back(c->workers[i]);
    const bool caller_multi = c->flood_multi;  // (lane 0 is the caller's own context: its setting comes back after the call)
    const bool caller_logs = c->flood_logs;
    const bool caller_jit = c->flood_jit;
    for (lr_context* l : lanes) {
        l->ransac_seed = c->ransac_seed;
        l->ransac_iters = c->ransac_iters;
        l->flood_mode = c->flood_mode;
        l->flood_staged = c->flood_staged;  // (off unless lr_set_flood_staged: +6 % in round 1, -3 % now)
        l->flood_partial = c->flood_partial;
        l->flood_giant_step = c->flood_giant_step;
        // Multi-source re-walks shorten a frame's rounds at the price of more work per long walk (eight wavefronts and a
        // second table entry per tile): worth it when the frame has the GPU to itself, not when S frames share it -- their
        // rounds overlap each other anyway (profiles/r04_flood_multi_sweep.txt).  LIBRECTIFY_FLOOD_MULTI_LANES=1 keeps them.
        static const bool multi_lanes = std::getenv("LIBRECTIFY_FLOOD_MULTI_LANES") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_MULTI_LANES")) != 0;
        l->flood_multi = caller_multi && (S == 1 || multi_lanes);
        // Re-walks from the logs: in the lanes only for walks of 32 tiles and more, behind a walk of 24 tiles.  With the
        // thresholds of a single call (16 / 12) round two's work on thousands of small logs is work on top, and S frames in
        // flight gain nothing from shorter rounds: 10.34 -> 10.2 Gpix/s; with these the long re-walks go and little is added:
        // 10.34 -> 10.55 (profiles/r04_flood_logs.txt section 14).  LIBRECTIFY_FLOOD_LOGS_LANES=0 keeps the lanes without.
        static const bool logs_lanes = !(std::getenv("LIBRECTIFY_FLOOD_LOGS_LANES") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOGS_LANES")) == 0);
        l->flood_logs = caller_logs && (S == 1 || logs_lanes);
        static const int lanes_min = std::getenv("LIBRECTIFY_FLOOD_LOG_MIN_LANES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOG_MIN_LANES")) : 32;
        static const int lanes_walk = std::getenv("LIBRECTIFY_FLOOD_LOG_WALK_LANES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOG_WALK_LANES")) : 24;
        l->flood_log_min = S == 1 ? 0 : lanes_min;
        l->flood_log_walk = S == 1 ? 0 : lanes_walk;
        static const int logs_lanes_from = std::getenv("LIBRECTIFY_FLOOD_LOGS_LANES_FROM") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOGS_LANES_FROM")) : 1;
        l->flood_log_from = S == 1 ? 1 : std::max(logs_lanes_from, 1);
        // ... and without the logs of second-tier walks: their kernel is a launch of 1 024 threads and 142 KB of LDS a
        // workgroup that has to find whole CUs beside the other lanes' kernels (2.7 launches a frame x 52 us in
        // profiles/r04_kernel_stats.csv): 10.67 -> 10.76 Gpix/s without (three repetitions each).  LIBRECTIFY_FLOOD_LOGBIG_LANES=1
        static const bool logbig_lanes = std::getenv("LIBRECTIFY_FLOOD_LOGBIG_LANES") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOGBIG_LANES")) != 0;
        l->flood_logbig_off = S > 1 && !logbig_lanes;
        l->flood_log_sweep = c->flood_log_sweep;
        // (a lane's thread has nothing else to do while its frame is in flight, but the call's staging threads need the cores:
        // a lane looks at the words every few tens of microseconds instead of spinning -- the other lanes keep the GPU busy)
        // Measured (profiles/r04_flood_logs.txt, section 9): 9.93 -> 10.29 Gpix/s from pageable frames, 11.7 -> 12.1 from
        // resident ones -- a frame of a lane no longer drags 3-4 rounds of empty launches through its stream.
        // LIBRECTIFY_FLOOD_JIT_LANES = microseconds between looks (20); 0 = blind rounds.
        static const int jit_lanes = std::getenv("LIBRECTIFY_FLOOD_JIT_LANES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_JIT_LANES")) : 20;
        l->flood_jit = caller_jit && (S == 1 || jit_lanes > 0);
        l->flood_jit_sleep_us = S == 1 ? 0 : jit_lanes;
        l->estimator = c->estimator;
        l->prosac_T_N = c->prosac_T_N;
        l->cht_d = c->cht_d;
        l->timing_on = c->timing_on;
        // LIBRECTIFY_LANES_SLEEP: the lanes sleep on an event instead of spinning in hipStreamSynchronize, for hosts
        // short of cores (on the 16-core share of a one-GPU box: pageable frames equal, page-locked ones 5 % slower)
        static const bool sleep_env = std::getenv("LIBRECTIFY_LANES_SLEEP") != nullptr;
        l->sleep_in_wait = sleep_env && h_frames != nullptr;
    }
    // ---- the upload ring
    const size_t npix = (size_t)w * h;
    int R = 0;
    bool any_pageable = false;
    if (h_frames) {
        if (w < 1 || h < 1) {
            set_error("upload: bad frame");
            return 1;
        }
        if ((stride < 0 ? -stride : stride) < w) {
            set_error("upload: |stride| smaller than the width");
            return 1;
        }
        for (int i = 0; i < batch; ++i) {
            if (h_frames[i] == nullptr) {
                set_error("upload: bad frame");
                return 1;
            }
        }
        for (int i = 0; i < batch && !any_pageable; ++i) any_pageable = !is_page_locked(h_frames[i]);
        // slots beyond one per lane: what the link may be ahead.  Six absorb the moments when several lanes finish
        // together (pageable 4K frames: 7.8 Gpix/s with three, 8.0 with six or ten); at most 2 GiB of frames, though.
        static const int extra_env = std::getenv("LIBRECTIFY_RING_EXTRA") ? std::max(1, std::atoi(std::getenv("LIBRECTIFY_RING_EXTRA"))) : 0;
        const int by_bytes = (int)std::min<size_t>(64, ((size_t)2 << 30) / (npix * sizeof(float)));
        R = extra_env ? S + extra_env : std::max(S + 1, std::min(S + 6, by_bytes));
        R = std::min(batch, R);
        if (ensure_upload_ring(c, R, npix, any_pageable)) return 1;
    }
    std::vector<std::atomic<int>> enq(h_frames ? (size_t)batch : 0);  // frame's slot + 1 once its transfer is enqueued
    std::vector<std::atomic<int>> slot_busy((size_t)R);
    for (auto& a : enq) a.store(0, std::memory_order_relaxed);
    for (auto& a : slot_busy) a.store(0, std::memory_order_relaxed);
    static const bool lane_debug = std::getenv("LIBRECTIFY_LANE_DEBUG") != nullptr;
    static const size_t batch_band = std::getenv("LIBRECTIFY_BATCH_BAND_KB") ? (size_t)std::max(64, std::atoi(std::getenv("LIBRECTIFY_BATCH_BAND_KB"))) << 10 : (size_t)4 << 20;
    std::atomic<int> abort_all{0};
    std::string up_err;
    auto nap = [](int& spins) {  // a wait that is usually short: yield first, then sleep
        if (++spins < 64) std::this_thread::yield();
        else std::this_thread::sleep_for(std::chrono::microseconds(30));
    };
    // LIBRECTIFY_BATCH_STATS: who waited for whom in this call (host clocks only, nothing is synchronised for it)
    static const bool batch_stats = std::getenv("LIBRECTIFY_BATCH_STATS") != nullptr;
    std::atomic<long long> up_wait_us{0}, lane_wait_us{0}, lead_sum_us{0};
    std::atomic<int> late_frames{0};
    // Pageable caller frames are page-locked WHERE THEY LIE (hipHostRegister) by a few helper threads running ahead of the
    // uploader, sent by DMA from there, and released when the call is over -- no staging copy, i.e. one pass through host DRAM
    // instead of three.  What it costs instead is the pinning itself (page-table work, ~1.3 ms a 4K frame on one thread).
    // Round 5: 10.93 -> 11.38 Gpix/s on the headline leg with four helpers, the 1080p batch unchanged;
    // the default since.  A frame whose registration is refused (memory somebody else has registered, a mapping that cannot be
    // pinned) goes through the staging copy as before; LIBRECTIFY_REGISTER_FRAMES=0 sends every frame that way,
    // =<n> sets the helpers (profiles/r05_h2d_register.txt).
    static const int register_threads = std::getenv("LIBRECTIFY_REGISTER_FRAMES") ? std::max(0, std::atoi(std::getenv("LIBRECTIFY_REGISTER_FRAMES"))) : 4;
    // (a host on which pinning is slow -- measured on this box: 1.3 ms a 4K frame on one thread, 25 GB/s; the guard trips below
    // 6 GB/s -- gets the staging copy back: for the rest of the call, and for the context's next 32 batch calls before it tries again)
    if (c->register_slow_calls > 0) --c->register_slow_calls;
    const bool reg_frames = register_threads > 0 && h_frames != nullptr && any_pageable && stride >= w && c->register_slow_calls == 0;
    std::atomic<int> reg_slow{0};
    std::vector<std::atomic<int>> reg_ready(reg_frames ? (size_t)batch : 0);
    for (auto& a : reg_ready) a.store(0, std::memory_order_relaxed);
    // reg_ready[i]: 0 nobody has touched the frame, 3 a helper is registering it, 1 registered, 2 the staging copy takes it
    // (page-locked already, registration refused, or the uploader got there first).  The helpers take the frames in order; a
    // frame the uploader reaches before any helper has is staged.  (Helpers that begin at the fourth frame, the uploader staging
    // the call's first three rather than waiting 1.3 ms for the first registration: 11.30 against 11.41 Gpix/s -- the staging
    // copies cost the helpers more than the wait costs the call; LIBRECTIFY_REGISTER_LEAD=<n> for the comparison.)
    static const int reg_lead = std::getenv("LIBRECTIFY_REGISTER_LEAD") ? std::max(0, std::atoi(std::getenv("LIBRECTIFY_REGISTER_LEAD"))) : 0;
    std::atomic<int> reg_next{std::min(batch, reg_lead)};
    std::vector<std::thread> reg_pool;
    if (reg_frames)
        for (int t = 0; t < register_threads; ++t)
            reg_pool.emplace_back([&]() {
                (void)hipSetDevice(c->device);
                bind_this_thread_near(c->device);
                for (int i = reg_next.fetch_add(1, std::memory_order_relaxed); i < batch && !abort_all.load(std::memory_order_relaxed);
                     i = reg_next.fetch_add(1, std::memory_order_relaxed)) {
                    if (reg_slow.load(std::memory_order_relaxed) >= 2) break;  // (pinning is slow here: the uploader stages the rest)
                    int expect = 0;
                    if (!reg_ready[(size_t)i].compare_exchange_strong(expect, 3, std::memory_order_acq_rel)) continue;  // (the uploader has it)
                    int ok = 2;
                    const double t_r0 = now_ms();
                    // (a frame that is in the batch more than once is registered by its first entry only: two helpers
                    // registering one range at the same time can both be told "done", and the second release then fails)
                    bool first = true;
                    for (int j = 0; j < i && first; ++j) first = h_frames[j] != h_frames[i];
                    if (first && !is_page_locked(h_frames[i]))
                        ok = hipHostRegister(const_cast<float*>(h_frames[i]), ((size_t)(h - 1) * (size_t)stride + (size_t)w) * sizeof(float), hipHostRegisterDefault) == hipSuccess ? 1 : 2;
                    if (ok == 2) (void)hipGetLastError();
                    if (ok == 1 && (now_ms() - t_r0) * 6.0e6 > (double)((size_t)h * (size_t)stride * sizeof(float))) reg_slow.fetch_add(1, std::memory_order_relaxed);
                    reg_ready[(size_t)i].store(ok, std::memory_order_release);
                }
            });
    const double t_call0 = now_ms();
    auto uploader = [&]() {
        if (hipSetDevice(c->device) != hipSuccess) {
            up_err = "hipSetDevice failed";
            abort_all.store(1);
            return;
        }
        if (any_pageable) bind_this_thread_near(c->device);  // (this thread stages too; it is the library's own)
        StagingCrew crew;
        if (any_pageable) crew.start(c, staging_threads(num_threads, npix * sizeof(float)) - 1);
        for (int i = 0; i < batch; ++i) {
            int slot = -1, spins = 0;
            const double t_w0 = batch_stats ? now_ms() : 0.0;
            // a slot whose frame is done (its transfer and its staging buffer are then free as well)
            while (slot < 0 && !abort_all.load(std::memory_order_relaxed)) {
                for (int k = 0; k < R && slot < 0; ++k)
                    if (!slot_busy[(size_t)k].load(std::memory_order_acquire)) slot = k;
                if (slot < 0) nap(spins);
            }
            if (batch_stats) up_wait_us.fetch_add((long long)((now_ms() - t_w0) * 1e3));
            if (abort_all.load(std::memory_order_relaxed)) return;
            slot_busy[(size_t)slot].store(1, std::memory_order_relaxed);
            if (reg_frames) {  // (the frame is being pinned by a helper: wait for it)
                int sp2 = 0;
                int expect = 0;
                if (!reg_ready[(size_t)i].compare_exchange_strong(expect, 2, std::memory_order_acq_rel))  // (else: nobody has it -- staged)
                    while (reg_ready[(size_t)i].load(std::memory_order_acquire) == 3 && !abort_all.load(std::memory_order_relaxed)) nap(sp2);
            }
            float* stage = is_page_locked(h_frames[i]) ? nullptr : c->ring_stage[(size_t)slot];
            const double t_u0 = now_ms();
            hipEvent_t e_dbg = nullptr;
            if (lane_debug && hipEventCreate(&e_dbg) == hipSuccess) (void)hipEventRecord(e_dbg, c->copy_stream);
            const int up_rc = stage ? crew.run(c->ring_img[(size_t)slot], stage, h_frames[i], w, h, stride, c->copy_stream, batch_band, batch_band > ((size_t)4 << 20) ? StagingCrew::kPieces : 1)
                                    : upload_rows(c, c->ring_img[(size_t)slot], nullptr, h_frames[i], w, h, stride, 1, c->copy_stream);
            if (up_rc || hipEventRecord(c->ring_ev[(size_t)slot], c->copy_stream) != hipSuccess) {
                up_err = get_error().empty() ? "upload failed" : get_error();
                abort_all.store(1);
                return;
            }
            enq[(size_t)i].store(slot + 1, std::memory_order_release);
            if (lane_debug) {
                const double t_u1 = now_ms();
                float dma = 0.f;  // from the moment the link was free for this frame to the end of its last transfer
                if (e_dbg) {
                    (void)hipEventSynchronize(c->ring_ev[(size_t)slot]);
                    if (hipEventElapsedTime(&dma, e_dbg, c->ring_ev[(size_t)slot]) != hipSuccess) (void)hipGetLastError();
                    (void)hipEventDestroy(e_dbg);
                }
                std::fprintf(stderr, "uploader frame %d: slot %d, waited %d naps for it, staged and enqueued in %.2f ms, on the link %.2f ms\n", i, slot, spins, t_u1 - t_u0, dma);
            }
        }
    };
    std::atomic<int> next_frame{0};
    std::vector<int> rc(S, 0);
    std::vector<std::string> err(S);
    auto work = [&](int si) {
        lr_context* l = lanes[si];
        auto fail = [&]() {
            rc[si] = 1;
            err[si] = get_error();
            abort_all.store(1);
        };
        if (si > 0 && hipSetDevice(c->device) != hipSuccess) {
            set_error("hipSetDevice failed");
            return fail();
        }
        for (int b = next_frame.fetch_add(1, std::memory_order_relaxed); b < batch; b = next_frame.fetch_add(1, std::memory_order_relaxed)) {
            const float* img = nullptr;
            int img_stride = stride, slot = -1;
            if (h_frames) {
                int spins = 0;
                const double t_w0 = batch_stats ? now_ms() : 0.0;
                while ((slot = enq[(size_t)b].load(std::memory_order_acquire) - 1) < 0 && !abort_all.load(std::memory_order_relaxed)) nap(spins);
                if (batch_stats) lane_wait_us.fetch_add((long long)((now_ms() - t_w0) * 1e3));
                if (slot < 0) return;  // (whoever stopped the batch has the message)
                if (hipStreamWaitEvent(l->stream, c->ring_ev[(size_t)slot], 0) != hipSuccess) {
                    set_error("hipStreamWaitEvent failed");
                    return fail();
                }
                img = c->ring_img[(size_t)slot];
                img_stride = w;
            } else {
                img = d_images + (size_t)b * image_stride;
            }
            std::vector<LineSegment> res;
            const double t_f0 = now_ms();
            if (ctx_find_groups_device(l, img, w, h, img_stride, min_length, refine, res)) return fail();
            if (batch_stats && h_frames && l->timing_on) {  // (with the stage timers on: ev[0] is the frame's first kernel)
                float lead = 0.f;
                if (hipEventElapsedTime(&lead, c->ring_ev[(size_t)slot], l->ev[0]) == hipSuccess) {
                    lead_sum_us.fetch_add((long long)(lead * 1e3));
                    if (lead < 0.05f) late_frames.fetch_add(1);
                } else {
                    (void)hipGetLastError();
                }
            }
            if (lane_debug) {
                float lead = 0.f;  // how long the frame's upload had been finished when its first kernel started
                if (h_frames && hipEventElapsedTime(&lead, c->ring_ev[(size_t)slot], l->ev[0]) != hipSuccess) (void)hipGetLastError();
                std::fprintf(stderr, "lane %d frame %d: enqueue %.2f ms, wait %.2f, whole call %.2f; device total %.2f; upload done %.2f ms before the first kernel\n",
                             si, b, l->host_ms[0], l->host_ms[2], now_ms() - t_f0, l->stage_ms[LR_T_TOTAL], lead);
            }
            if (h_frames) slot_busy[(size_t)slot].store(0, std::memory_order_release);
            const int n = (int)res.size();
            if (n_lines) n_lines[b] = n;
            if (out && capacity > 0)
                std::memcpy(out + (size_t)b * capacity, res.data(), sizeof(LineSegment) * (size_t)std::min(n, capacity));
            if (transforms) {
                const RectificationConfig def;
                transforms[b] = rectification_transform(res.data(), std::min(n, capacity > 0 ? capacity : n), w, h,
                                                        cfg ? *cfg : def);
            }
        }
    };
    std::vector<std::thread> th;
    if (h_frames) th.emplace_back(uploader);
    for (int si = 1; si < S; ++si) th.emplace_back(work, si);
    work(0);
    for (auto& t : th) t.join();
    if (batch_stats && h_frames)
        std::fprintf(stderr, "batch of %d frames: %.2f ms; the uploader waited %.2f ms for free device frames; the %d lanes waited %.2f ms in all for their frame's transfer to be enqueued\n",
                     batch, now_ms() - t_call0, up_wait_us.load() * 1e-3, S, lane_wait_us.load() * 1e-3);
    if (batch_stats && h_frames && c->timing_on)
        std::fprintf(stderr, "   a frame's transfer was finished %.3f ms (mean) before its first kernel started; %d of %d frames started within 0.05 ms of it (they waited for the link)\n",
                     lead_sum_us.load() * 1e-3 / batch, late_frames.load(), batch);
    if (h_frames) (void)hipStreamSynchronize(c->copy_stream);  // (after an error: nothing may still read the caller's frames)
    for (auto& t : reg_pool) t.join();
    if (reg_slow.load() >= 2) c->register_slow_calls = 32;
    if (reg_frames)  // the caller's frames go back to being pageable
        for (int i = 0; i < batch; ++i)
            if (reg_ready[(size_t)i].load(std::memory_order_acquire) == 1 && hipHostUnregister(const_cast<float*>(h_frames[i])) != hipSuccess)
                (void)hipGetLastError();  // (nothing to be done about it, and the caller's thread must not find it in its next call)
    for (lr_context* l : lanes) l->sleep_in_wait = false;
    c->flood_multi = caller_multi;
    c->flood_logs = caller_logs;
    c->flood_log_from = 1;
    c->flood_logbig_off = false;
    c->flood_log_min = c->flood_log_walk = 0;
    c->flood_jit = caller_jit;
    c->flood_jit_sleep_us = 0;
    for (int si = 0; si < S; ++si)
        if (rc[si]) {
            set_error(err[si]);
            return 1;
        }
    if (abort_all.load()) {
        set_error(up_err.empty() ? "batch: upload failed" : up_err);
        return 1;
    }
    return 0;
}

int ctx_find_groups_batch_device(lr_context* c, const float* d_images, size_t image_stride, int batch, int w, int h,
                                 int stride, float min_length, bool refine, LineSegment* out, int capacity, int* n_lines,
                                 const RectificationConfig* cfg, ImageTransform* transforms) {
    return find_groups_batch(c, d_images, image_stride, nullptr, batch, w, h, stride, min_length, refine, -1, out,
                             capacity, n_lines, cfg, transforms);
}

int ctx_find_groups_batch_host(lr_context* c, const float* const* frames, int batch, int w, int h, int stride,
                               float min_length, bool refine, int num_threads, LineSegment* out, int capacity,
                               int* n_lines, const RectificationConfig* cfg, ImageTransform* transforms) {
    return find_groups_batch(c, nullptr, 0, frames, batch, w, h, stride, min_length, refine, num_threads, out, capacity,
                             n_lines, cfg, transforms);
}

// One batch call over several devices of this process.  Every entry of the device list has a context of its own (kept
// with `c` from call to call), configured like `c`, with its own uploader, copy stream, lanes and pool; a host thread per
// entry runs the single-device batch call on its contiguous block of the frames.
int ctx_find_groups_batch_host_multi(lr_context* c, const int* devices, int n_devices, const float* const* frames, int batch,
                                     int w, int h, int stride, float min_length, bool refine, int num_threads,
                                     LineSegment* out, int capacity, int* n_lines, const RectificationConfig* cfg,
                                     ImageTransform* transforms) {
    if (batch <= 0) return 0;
    if (n_devices <= 0 || devices == nullptr) {
        set_error("multi-device batch: empty device list");
        return 1;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) ndev = 0;
    for (int i = 0; i < n_devices; ++i)
        if (devices[i] < 0 || devices[i] >= ndev) {
            set_error("multi-device batch: device index out of range");
            return 1;
        }
    // the contexts of the list (re-made where the list changed)
    if ((int)c->peers.size() > n_devices) {
        for (size_t i = (size_t)n_devices; i < c->peers.size(); ++i) ctx_destroy(c->peers[i]);
        c->peers.resize((size_t)n_devices);
    }
    c->peers.resize((size_t)n_devices, nullptr);
    for (int i = 0; i < n_devices; ++i) {
        if (c->peers[(size_t)i] && c->peers[(size_t)i]->device != devices[i]) {
            ctx_destroy(c->peers[(size_t)i]);
            c->peers[(size_t)i] = nullptr;
        }
        if (!c->peers[(size_t)i] && ctx_create(devices[i], &c->peers[(size_t)i])) {
            (void)hipSetDevice(c->device);
            return 1;
        }
        lr_context* p = c->peers[(size_t)i];
        p->ransac_seed = c->ransac_seed;
        p->ransac_iters = c->ransac_iters;
        p->flood_mode = c->flood_mode;
        p->flood_staged = c->flood_staged;
        p->flood_partial = c->flood_partial;
        p->flood_multi = c->flood_multi;
        p->flood_logs = c->flood_logs;
        p->flood_log_sweep = c->flood_log_sweep;
        p->flood_giant_step = c->flood_giant_step;
        p->flood_jit = c->flood_jit;
        p->flood_aux_on = c->flood_aux_on;
        p->seed_keep_ratio = c->seed_keep_ratio;
        p->estimator = c->estimator;
        p->prosac_T_N = c->prosac_T_N;
        p->cht_d = c->cht_d;
        p->timing_on = c->timing_on;
        p->batch_streams = c->batch_streams;
    }
    const int per = (batch + n_devices - 1) / n_devices;  // contiguous blocks of ceil(B / G) frames
    std::vector<int> rc((size_t)n_devices, 0);
    std::vector<std::string> err((size_t)n_devices);
    auto run = [&](int i) {
        const int b0 = std::min(batch, i * per), b1 = std::min(batch, b0 + per);
        if (b1 <= b0) return;
        lr_context* p = c->peers[(size_t)i];
        if (hipSetDevice(p->device) != hipSuccess) {
            rc[(size_t)i] = 1;
            err[(size_t)i] = "hipSetDevice failed";
            return;
        }
        // (a share of the staging threads each: they all copy out of the same host memory)
        const int nt = num_threads > 1 ? std::max(2, num_threads / std::min(n_devices, (batch + per - 1) / per)) : num_threads;
        if (ctx_find_groups_batch_host(p, frames + b0, b1 - b0, w, h, stride, min_length, refine, nt,
                                       out ? out + (size_t)b0 * capacity : nullptr, capacity, n_lines ? n_lines + b0 : nullptr, cfg,
                                       transforms ? transforms + b0 : nullptr)) {
            rc[(size_t)i] = 1;
            err[(size_t)i] = get_error();  // (thread-local: carried back to the caller below)
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < n_devices; ++i) th.emplace_back(run, i);
    run(0);
    for (auto& t : th) t.join();
    (void)hipSetDevice(c->device);
    for (int i = 0; i < n_devices; ++i)
        if (rc[(size_t)i]) {
            set_error("device list entry " + std::to_string(i) + " (device " + std::to_string(devices[i]) + "): " + err[(size_t)i]);
            return 1;
        }
    return 0;
}

}  // namespace lramd
