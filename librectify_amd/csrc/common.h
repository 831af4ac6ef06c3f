// Internal declarations shared by the HIP kernels and the host pipeline.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>

#include "../../include/librectify_amd.h"

namespace lramd {

// reference config.h:23-59
constexpr float kEps = 1e-6f;
constexpr int kMaxModels = 4;
constexpr int kMaxPeelModels = 5;  // peeling rounds the device buffers are sized for (d_models: words 16.. are timing slots)
constexpr float kInlierDeg = 2.0f;
constexpr float kGarbageDeg = 4.0f;
constexpr int kRansacMaxIter = 10000;
constexpr int kEdgeKernelSize = 2;
constexpr float kEdgeKernelSigma = 1.0f;
constexpr int kSeedDist = 2;
constexpr float kSeedRatio = 0.95f;
constexpr float kTraceTolerance = 0.25f;
constexpr float kLineMaxErr = 2.0f;
constexpr float kLineMinLength = 5.f;
constexpr int kComponentMinSize = 5;
constexpr int kBins = 8;

struct FilterConsts {
    float d[5];  // 1-D factors of the 5x5 taps (reference filter.cpp:65-78): Hx(i,j) = d[j] g[i], Hy(i,j) = d[i] g[j]
    float g[5];
    float st[kBins];  // sin(theta_b), theta_b = float(b*pi)/8 (line_detector.cpp:144-145)
    float ct[kBins];  // cos(theta_b)
};

struct BinTrig {
    float st[kBins];
    float ct[kBins];
};

constexpr uint32_t kLabelFree = 0xFFFFFFFFu;

// Directional edge response of bin b (line_detector.cpp:145), canonical form.
__host__ __device__ inline float directional(float dx, float dy, float s, float c) {
    return fabsf(fmaf(dx, s, dy * c));
}

// Counter-based RANSAC sample generator: a uniform sorted pair (a < b) out of n, a pure function
// of (seed, round, iteration).  Replaces choice_knuth under mt19937(random_device)
// (reference estimator.h:35,49-50, math_utils.cpp:14-39): same distribution, no sequential state.
__host__ __device__ inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__host__ __device__ inline void sample_pair(uint64_t seed, uint32_t round, uint32_t iter, uint32_t n, uint32_t& a,
                                            uint32_t& b) {
    const uint64_t z = splitmix64(seed ^ splitmix64(((uint64_t)round << 32) | iter));
    const uint32_t u1 = (uint32_t)z, u2 = (uint32_t)(z >> 32);
    const uint32_t i = (uint32_t)(((uint64_t)u1 * n) >> 32);
    uint32_t j = (uint32_t)(((uint64_t)u2 * (n - 1)) >> 32);
    if (j >= i) ++j;
    a = i < j ? i : j;
    b = i < j ? j : i;
}

void set_error(const std::string& msg);

#define LR_HIP(expr)                                                                                     \
    do {                                                                                                 \
        hipError_t _e = (expr);                                                                          \
        if (_e != hipSuccess) {                                                                          \
            (void)hipGetLastError(); /* reported here: must not resurface behind a later launch */       \
            ::lramd::set_error(std::string(#expr) + ": " + hipGetErrorString(_e) + " (" + __FILE__ + ":" + \
                               std::to_string(__LINE__) + ")");                                          \
            return 1;                                                                                    \
        }                                                                                                \
    } while (0)

// ---- launchers (each enqueues on `s`, never synchronises unless stated) -----------------
// kernels_filter.hip
int launch_filter(const float* img, int w, int h, int stride, const FilterConsts& fc, float* dx, float* dy,
                  uint8_t* dmask, uint64_t* cand, uint32_t* cand_count, uint32_t* tile_max, hipStream_t s);
int filter_band_rows();  // image rows per band of the filter kernel
int filter_band_last_row(int by);  // last image row band row `by` reads (its first is filter_band_rows() * by - 4)
int launch_filter_rows(const float* img, int w, int h, int stride, const FilterConsts& fc, float* dx, float* dy,
                       uint8_t* dmask, uint64_t* cand, uint32_t* cand_count, uint32_t* tile_max, int by_begin, int by_end,
                       hipStream_t s);
struct FilterGeom {
    int n_tiles;   // candidate lists written by the filter (tiles or bands, by kernel variant)
    int cand_cap;  // slots per list
};
FilterGeom filter_geometry(int w, int h);

// kernels_seeds.hip
size_t seeds_temp_bytes(int n_tiles, size_t max_seeds);
int launch_seed_select(const uint64_t* cand, const uint32_t* cand_count, const uint32_t* tile_max, int n_tiles,
                       int cand_cap, float seed_keep_ratio, float* maxmag, uint32_t* tile_pass, uint32_t* tile_off, uint64_t* keys,
                       uint32_t key_cap, uint32_t* n_seeds, uint32_t frame_tag, hipStream_t s);
// the seed count stays on the device (*n_seeds, clamped to cap); the launches cover `cap` seeds.  keys_alt: a second buffer
// of cap keys (the merge rounds of large frames go back and forth between the two)
int launch_seed_order(uint64_t* keys, uint64_t* keys_alt, const uint32_t* n_seeds, uint32_t cap, const float* dx, const float* dy,
                      BinTrig trig, float trace_tolerance, int32_t* seed_idx, int32_t* seed_bin, float* seed_thr, hipStream_t s);

// kernels_flood.hip
int launch_label_init(uint32_t* label, size_t n, hipStream_t s);
int launch_flood_ordered(const float* dx, const float* dy, const uint8_t* dmask, int w, int h, const int32_t* seed_idx,
                         const int32_t* seed_bin, const float* seed_thr, const uint32_t* d_n_seeds, uint32_t seed_cap,
                         BinTrig trig, uint32_t* label, int32_t* seed_size, int32_t* queue, hipStream_t s);

// parallel-round flood (mode 1): per-seed round state, active lists and overflow slabs
constexpr size_t kFloodWpWords = 8;      // per seed: header (way-points | tiles of the walk << 8) and seven pixel indices
constexpr size_t kFloodHandWords = 1024;  // per hand-over: header, up to 192 table entries, up to 128 frontier records
struct FloodBuffers {
    uint32_t* blocked = nullptr;
    uint32_t* count = nullptr;
    uint32_t* flags = nullptr;
    uint8_t* state = nullptr;
    uint8_t* tier = nullptr;        // per seed: 1 = its walk outgrew the first storage tier in an earlier round
    uint32_t* blk = nullptr;        // per seed: the lower seed that blocked its last long walk (kernels_flood.hip: kCtrlDeferLow)
    uint32_t* act_a = nullptr;
    uint32_t* act_b = nullptr;
    uint32_t* ctrl = nullptr;       // kFloodCtrlWords words
    uint8_t* dirty = nullptr;       // one mark per 256 pixels of the label image: stamped in the current round
    uint32_t* big_list = nullptr;   // 8192 seeds: this round's hand-over to the second storage tier
    uint32_t* handover = nullptr;   // ... and the state each of their walks had reached (kFloodHandWords words per list entry)
    // Way-points of long walks (round 4): a seed whose walk covered many tiles leaves kFloodWpWords - 1 pixels spread over its
    // footprint; if it has to walk again, a team of wavefronts starts from the seed AND from those pixels at once
    // (kernels_flood.hip: team_walk, kMulti).  One record per seed for the first wp_cap seeds.
    uint32_t* waypoints = nullptr;
    uint32_t wp_cap = 0;
    bool multi_source = true;       // lr_set_flood_multi_source / LIBRECTIFY_FLOOD_MULTI=0: comparison hook, same labels
    uint32_t* multi_list = nullptr; // way-point seeds of the coming round (8192): walked by a team launch beside the exploration ...
    hipStream_t aux_stream = nullptr;          // ... on this second stream of the context (nullptr: they go through big_list, after it)
    const hipEvent_t* fork_events = nullptr;   // one pair per round that forks (enqueue_round)
    const hipEvent_t* join_events = nullptr;
    int n_fork_events = 0;
    int multi_round_last = 4;       // last round (index from 0) that walks its way-point seeds beside its exploration
    // Re-walks from the log (round 4; kernels_flood.hip: flood_rewalk_kernel).  A blocked seed walks its footprint again
    // every round, tile after tile, and rounds 2-5 of a frame last as long as their one longest such walk.  But a footprint
    // only ever SHRINKS (acceptance is static but for commits), so the next footprint is the connected part around the seed
    // of (last footprint minus committed pixels): a finished walk leaves its (tile, pixels) records here, and the later
    // rounds label the components of those records in LDS -- no dependent chain of memory round trips.
    bool rewalk_logs = false;
    int log_min_tiles = 0, log_walk_tiles = 0;  // 0: the defaults (16 and 12, LIBRECTIFY_FLOOD_LOG_MIN / _WALK); the lanes of a batch bring their own
    int log_from_round = 1;    // first round (from 0) whose seeds turn to their logs (walks leave logs from the first round on)
    bool giant_hold = false;   // only the lowest active seed walks on into a global slab; other walks that outgrow the second tier are held back (kernels_flood.hip: kCtrlLowest)
    // The giant step (round 5; kernels_flood.hip: kCtrlGiantStep): when the lowest active seed's walk outgrows the LDS tiers, its
    // flood -- a plain connected component, nothing speculative about it -- is labelled by the whole device between two
    // rounds (tile masks + union-find) instead of being walked by one team of wavefronts through a slab.
    bool giant_step = true;          // LIBRECTIFY_FLOOD_GIANT_STEP=0: the slab walk, as before (comparison)
    void* giant_mask = nullptr;      // one 64-bit mask per 8x8 tile of the frame
    uint32_t* giant_parent = nullptr;  // one word per pixel (the ordered kernels' queue: never in use at the same time)
    bool log_sweep = false;    // test hook: the fall-back (sweeps) for every log
    bool rewalk_big = false;   // the frame is expected to have walks beyond the first tier: their logs are kept too, and a second launch per round works on them
    uint32_t log_seeds = 0, log_cap = 0;
    uint32_t* log_off = nullptr;
    uint32_t* log_len = nullptr;
    uint32_t* log_buf = nullptr;
    void* slab_ring = nullptr;  // n_slabs x slab_ring_cap 16-byte records
    void* slab_hash = nullptr;  // n_slabs x slab_hash_cap 16-byte records
    uint32_t n_slabs = 0, slab_ring_cap = 0, slab_hash_cap = 0;
    // Staged start of the rounds: the first round walks the strongest n_seeds >> win_first_shift seeds (0 = all),
    // the window grows by << win_growth per round.  Fewer pixels are walked in total (weak seeds on an edge that a
    // strong seed takes die unwalked) at the price of one or two more rounds: better throughput with many frames
    // in flight, worse latency for a single frame.
    int win_first_shift = 0, win_growth = 2;
    // Hold-back: on frames whose first full round shows walks beyond the first storage tier (four or more), the weakest
    // (100 - pct) % of the seeds wait until all others are resolved (0 = never).  Measured at 4K with pct = 80: a
    // natural image 8.6 -> ~5 ms, the long-edge stress frame 11 -> ~6 ms, frames without such walks untouched.
    int win_hold_pct = 80;
    // Partial commits (kernels_flood.hip: flood_partial_commit_kernel): blocked seeds commit at once the part of their
    // footprint that no lower seed can reach.  LIBRECTIFY_FLOOD_PARTIAL=0 switches them off (comparison).
    bool partial_commits = true;
    bool second_tier = true;  // test hook: without it every walk that outgrows the first tier goes to a slab
    // Whether a flood starts with the second tier or only turns it on once a walk has had to go to a slab.  Since round 3
    // the context always asks for it from the start (context.hip, finish_flood: a frame of regions after a frame of lines
    // otherwise runs its long walks in slabs, 6.4 instead of 1.7 ms of flood at 4K; an empty launch costs a round 5 us).
    bool second_tier_from_start = true;
    // Likewise the hold-back: if the context's previous frame engaged it, this frame starts with it (a round of very
    // long walks saved); otherwise it engages after the first full round that shows such walks.
    bool hold_from_start = false;
    // ... and the staged window of a frame of overlapping giants (kernels_flood.hip: kCtrlStaged): a frame that holds back
    // many walks in its first round goes on with the strongest quarter of its seeds; the frame after it starts that way.
    bool staged_from_start = false;
    int blind_rounds = 10;  // rounds flood_enqueue enqueues without looking at the control block
    // Rounds just in time (round 4; single calls): the first `jit_first` rounds are enqueued blindly, every further one only
    // when the host has SEEN, in page-locked words the last workgroup of a round writes, that the rounds so far left seeds
    // -- no launch behind the last round with work (the blind rounds of a 4K frame were 60-120 us of empty launches), at
    // the price of the host's reaction time per further round.  The calling thread polls while the flood runs.
    uint32_t* host_progress = nullptr;  // page-locked, device-visible, 8-byte aligned: one 64-bit report (kernels_flood.hip: flood_report)
    bool calm_hint = false;             // the context's last frame had no walk beyond the first tier: blind rounds 2.. go without the second tier's launch
    uint32_t* host_ctrl = nullptr;      // page-locked, device-visible: the round that ends the flood leaves the control block here (no copy of it is enqueued then)
    int jit_first = 0;                  // 0: off
    int jit_sleep_us = 0;               // the polling thread sleeps this long between looks (0: it spins -- single calls)
    int jit_lead = 0;                   // rounds the host keeps enqueued ahead of the last one it has seen finished
    uint32_t big_cap_override = 0;  // test hook: seeds per round the second tier takes (0 = the default, 8192)
    uint32_t team_tile_cap = 0;     // test hook: tiles after which the second tier's team hands a walk to a slab (0 = its table)
};
// What a frame hands to the flood.  The seed count stays on the device (*d_n_seeds, clamped to seed_cap, the
// capacity the seed sort ran with): launches are sized by seed_cap.
struct FloodFrame {
    const float* dx;
    const float* dy;
    const uint8_t* dmask;
    int w, h;
    const int32_t* seed_idx;
    const int32_t* seed_bin;
    const float* seed_thr;
    const uint32_t* d_n_seeds;
    uint32_t seed_cap;
    BinTrig trig;
    uint32_t* label;
    int32_t* seed_size;
    int32_t* queue;
};
struct FloodProgress {
    int enqueued = 0;  // rounds enqueued so far
    bool use_big = false;
    int win_growth = 2;
    // what the host that enqueued the rounds just in time has seen when the flood is over: the largest flood committed
    // (sizes_known = false: blind rounds, or a flood that flood_finish had to complete; max_flood: 0 or "more than 2^14 pixels")
    bool sizes_known = false;
    uint32_t max_flood = 0;
};
constexpr int kFloodCtrlWords = 56;
// Enqueues the initialisation and a first batch of rounds, then an asynchronous copy of the control block into
// h_ctrl (kFloodCtrlWords words of pinned host memory).  Never synchronises (except in LIBRECTIFY_FLOOD_DEBUG mode).
int flood_enqueue(const FloodBuffers& B, const FloodFrame& F, FloodProgress* P, uint32_t* h_ctrl, hipStream_t s);
// After the stream has been synchronised: completes the flood if the first batch did not (more rounds, ordered tail;
// synchronises).  *extra = the label image changed after flood_enqueue's rounds, so later stages must run again.
int flood_finish(const FloodBuffers& B, const FloodFrame& F, FloodProgress* P, uint32_t* h_ctrl, int* rounds_out,
                 uint32_t* tiers_out /* [14]: seeds moved to the second tier, slabs used, seeds left to the ordered tail,
                                        hold-back engaged, pixels walked (lo, hi), tile steps (lo, hi), ... giant steps */,
                 bool* extra, hipStream_t s);

// kernels_fit.hip (all counts stay on the device: launches cover seed_cap / comp_cap)
size_t fit_temp_bytes(size_t max_pixels, uint32_t max_segments);
// Pixel lists of more than 2^14 pixels ("huge" components) are dealt into buckets of 2^14 consecutive pixel indices and
// sorted bucket by bucket (kernels_fit.hip: huge_count_kernel, huge_sort_kernel).  tab: max x ceil(pixels / 2^14) words,
// ZERO between frames (the sort leaves them so); list: max words; jobs: four words per word of tab.
struct HugeSort {
    uint32_t* tab = nullptr;
    uint32_t* list = nullptr;
    uint32_t* jobs = nullptr;
    uint32_t max = 0;
};
// n_large: seven words (kernels_fit.hip), cleared by the launch; cursor: one word per component, written by it;
// temp: fit_temp_bytes() bytes, ZERO when allocated (tagged status words); frame_tag: not 0, different from call to call
int launch_component_offsets(const int32_t* seed_size, const uint32_t* d_n_seeds, uint32_t seed_cap, int min_size,
                             uint32_t* comp_rank, uint32_t* comp_seed, uint32_t* comp_off,
                             uint32_t* totals /*[0]=n_comp,[1]=n_px*/, uint32_t* large_list, uint32_t large_cap,
                             uint32_t* n_large, void* temp, size_t temp_bytes, uint32_t frame_tag, uint32_t* cursor, const HugeSort& hs,
                             hipStream_t s);
// with_huge = false: the caller knows that no flood of the frame has more than 2^14 pixels (the launches for those are left out)
int launch_component_scatter(const uint32_t* label, size_t npix, const uint32_t* comp_rank, const uint32_t* comp_off,
                             uint32_t* cursor, uint32_t* px, const HugeSort& hs, uint32_t* n_large, bool with_huge, hipStream_t s);
// scratch: >= 2 x (pixels of the frame) words, used by lists of more than 2^14 pixels that found no row in the huge table only
int launch_component_sort(const uint32_t* px_in, uint32_t* px_out, const uint32_t* comp_off, const uint32_t* d_n_comp,
                          uint32_t comp_cap, const uint32_t* large_list, uint32_t large_cap, const uint32_t* n_large,
                          uint32_t* scratch, const uint32_t* cursor, const HugeSort& hs, bool with_huge, hipStream_t s);
int launch_fit(const uint32_t* px_sorted, const uint32_t* px_unsorted, const uint32_t* comp_off, const uint32_t* comp_seed, const uint32_t* d_n_comp,
               uint32_t comp_cap, const int32_t* seed_bin, const float* dx, const float* dy, int w, BinTrig trig,
               float* scratch_w, LineSegment* out, const uint32_t* cursor, const HugeSort& hs, const uint32_t* n_large, bool with_huge,
               hipStream_t s);

// kernels_ransac.hip
struct PencilSoA {  // device pointers, n entries each (lines of the current round, compacted)
    const float* ax;
    const float* ay;
    const float* dx;
    const float* dy;
    const float* len;
    const float* hx;
    const float* hy;
    const float* hz;
};
constexpr int kRansacBestSlots = 32;  // 64-bit words the scoring workgroups reduce their best (score, iteration) into
// one solve (scoring launch + read-out): best score bits and iteration arrive in host_best[0..1] (page-locked); best_slots:
// zeroed device words (kRansacBestSlots x 8 bytes), left zeroed
int launch_ransac_score(PencilSoA m, uint32_t n, float tol, float degeneracy_tol, uint32_t n_iter, uint64_t seed,
                        uint32_t round, unsigned long long* best_slots, uint32_t* host_best, hipStream_t s);
// peeling rounds: the best lands in best_slots, which the round's peel launch reads and clears
int launch_ransac_score_dev(PencilSoA m, const uint32_t* gctl, int max_models, float tol, float degeneracy_tol,
                            uint32_t n_iter, uint64_t seed, unsigned long long* best_slots, hipStream_t s);

// kernels_groups.hip: filter_lines and the vanishing-point peeling on the device
struct PencilTable {  // PencilSoA plus the index of each entry in the frame's list of filtered lines
    float* ax;
    float* ay;
    float* dx;
    float* dy;
    float* len;
    float* hx;
    float* hy;
    float* hz;
    uint32_t* orig;
    PencilSoA soa() const { return PencilSoA{ax, ay, dx, dy, len, hx, hy, hz}; }
};
enum {  // words of the peeling control block
    kGcLines = 0,      // filtered lines of the frame (the model's size)
    kGcRemaining = 1,  // lines neither grouped nor garbage yet
    kGcRound = 2,      // peeling rounds done
    kGcActive = 3,     // lines in the compacted table of the coming round (== kGcRemaining)
    kGcBestIter0 = 4,  // [4..7]: winning iteration of each round (diagnostics)
    kGcWords = 8,
};
// all / round0 (both or neither): the kept segments' pencil model goes into these tables in the same launch (launch_pencil_model's work)
int launch_filter_lines(const LineSegment* raw, const uint32_t* d_n_raw, uint32_t raw_cap, float min_length,
                        LineSegment* out, uint32_t* gctl, float* gnorm, const PencilTable* all, const PencilTable* round0,
                        hipStream_t s);
int launch_lines_bbox(LineSegment* lines, uint32_t n, uint32_t* gctl, float* gnorm, hipStream_t s);
int launch_pencil_model(const LineSegment* lines, const uint32_t* gctl, const float* gnorm, PencilTable all,
                        PencilTable round0, uint32_t line_cap, hipStream_t s);
int launch_peel(PencilTable cur, PencilTable nxt, PencilTable all, unsigned long long* best_slots, uint64_t seed,
                float tol, float garbage_tol, int max_models, uint32_t* gctl, float* stage4 /* 4 floats per line */,
                LineSegment* lines, float* models,
                const uint32_t* counts, uint32_t cap_lines, void* gather_block /* not nullptr: the frame's result block is written at the end */,
                hipStream_t s);
int launch_cht_accumulate(PencilSoA m, uint32_t n, int d, unsigned long long* acc, hipStream_t s);
int launch_cht_votes(PencilSoA m, const uint32_t* idx, uint32_t n, int d, unsigned long long* acc, bool subtract,
                     unsigned long long* n_votes, hipStream_t s);
int launch_cht_peak(const unsigned long long* acc, int d, uint32_t* out3, hipStream_t s);
// refine: seg = n records of 7 floats {x1,y1,x2,y2,dx,dy,len}; edges = pairs of uint32 (i<j); *n_edges may exceed cap
int launch_refine_pairs(const void* seg, uint32_t n, void* edges, uint32_t* n_edges, uint32_t cap, hipStream_t s);
int launch_prosac_count(PencilSoA m, uint32_t n, float tol, float degeneracy_tol, const uint32_t* sa,
                        const uint32_t* sb, uint32_t n_hyp, uint32_t* counts, hipStream_t s);
int launch_prosac_flags(PencilSoA m, uint32_t n, float px, float py, float pz, float tol, uint8_t* flags, hipStream_t s);
// new-best iterations of a chunk (from its counts and the best count it starts with) and their inlier flags
int launch_prosac_records(PencilSoA m, uint32_t n, const uint32_t* sa, const uint32_t* sb, const uint32_t* counts,
                          uint32_t n_hyp, uint32_t best_in, float tol, uint32_t* rec, uint32_t cap, uint8_t* flags, hipStream_t s);
int launch_ht_weights(PencilSoA m, uint32_t n, const int32_t* pa, const int32_t* pb, int n_pairs, int ht, float* peak,
                      float* weights, hipStream_t s);

}  // namespace lramd
