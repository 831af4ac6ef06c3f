// Stage 3 — ordered flood (reference filter.cpp:101-153 + line_detector.cpp:92-122).
//
// Semantics to reproduce exactly: seeds are visited in descending-magnitude order; a seed
// whose pixel is already claimed is skipped; otherwise its flood claims the 8-connected set
// of pixels around it that are not yet claimed by ANY earlier flood (floods later discarded
// as too small included) and whose response in the seed's direction bin exceeds
// (1-TRACE_TOLERANCE) * response(seed).  The result is the label image
//      label[p] = index of the seed whose flood claimed p, or kLabelFree
// from which stage 4 derives the components (size > COMPONENT_MIN_SIZE).
//
// The eight masked directional planes are never stored; the acceptance test evaluates
//      (dmask[q] >> bin) & 1  &&  |fmaf(dx[q], sin_bin, dy[q]*cos_bin)| > thr
// on the fly (dmask = 0 on the 1-px border, which also stands in for flood_init_mask).
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <sys/prctl.h>
#include <vector>

#include "common.h"

namespace lramd {
namespace {

__device__ inline uint32_t ld_agent(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void label_init_kernel(uint32_t* __restrict__ label, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t step = (size_t)gridDim.x * 256;
    for (; i < n; i += step) label[i] = kLabelFree;
}

// Mode 0: one wavefront walks the seeds in order.  Each flood is a wave-parallel frontier
// expansion: 8 frontier pixels x 8 neighbours per step, claims by atomicCAS, ballot-compacted
// pushes.  The queue lives in one global array; because floods run one after another and each
// pixel is claimed once, seed k's queue segment is exactly its component.
__global__ __launch_bounds__(64) void flood_ordered_kernel(const float* __restrict__ dx, const float* __restrict__ dy,
                                                           const uint8_t* __restrict__ dmask, int w,
                                                           const int32_t* __restrict__ seed_idx,
                                                           const int32_t* __restrict__ seed_bin,
                                                           const float* __restrict__ seed_thr,
                                                           const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                           BinTrig trig, uint32_t* label, int32_t* seed_size,
                                                           int32_t* queue) {
    const uint32_t n_seeds = min(*n_ptr, cap);
    const int lane = threadIdx.x;
    const int si = lane >> 3, ni = lane & 7;
    // neighbour order of filter.cpp:130-137 (only the set matters)
    const int dr = (ni == 2 || ni == 6 || ni == 7) ? 1 : ((ni == 3 || ni == 4 || ni == 5) ? -1 : 0);
    const int dc = (ni == 0 || ni == 4 || ni == 6) ? -1 : ((ni == 1 || ni == 5 || ni == 7) ? 1 : 0);
    const int noff = dr * w + dc;
    size_t base = 0;
    for (uint32_t k = 0; k < n_seeds; ++k) {
        const int sidx = seed_idx[k];
        int size = 0;
        if (ld_agent(&label[sidx]) == kLabelFree) {
            const int b = seed_bin[k];
            const float thr = seed_thr[k];
            const float s = trig.st[b], c = trig.ct[b];
            const bool seed_ok = ((dmask[sidx] >> b) & 1) && (directional(dx[sidx], dy[sidx], s, c) > thr);
            if (seed_ok) {
                if (lane == 0) {
                    atomicExch(&label[sidx], k);
                    __hip_atomic_store(&queue[base], sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int head = 0, tail = 1;
                while (head < tail) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");  // queue traffic is sc1 (L2); this only orders it
                    const int nsrc = min(8, tail - head);
                    bool claim = false;
                    int q = 0;
                    if (si < nsrc) {
                        const int p = __hip_atomic_load(&queue[base + head + si], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q = p + noff;  // p is never on the image border (dmask = 0 there), so q is inside
                        if ((dmask[q] >> b) & 1) {
                            if (directional(dx[q], dy[q], s, c) > thr) claim = atomicCAS(&label[q], kLabelFree, k) == kLabelFree;
                        }
                    }
                    const uint64_t m = __ballot(claim);
                    if (claim) {
                        const int rank = __popcll(m & ((1ull << lane) - 1ull));
                        __hip_atomic_store(&queue[base + tail + rank], q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    tail += (int)__popcll(m);
                    head += nsrc;
                }
                size = tail;
                base += (size_t)tail;
            }
        }
        if (lane == 0) seed_size[k] = size;
    }
}

}  // namespace

int launch_label_init(uint32_t* label, size_t n, hipStream_t s) {
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(label_init_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, label, n);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_flood_ordered(const float* dx, const float* dy, const uint8_t* dmask, int w, int h, const int32_t* seed_idx,
                         const int32_t* seed_bin, const float* seed_thr, const uint32_t* d_n_seeds, uint32_t seed_cap,
                         BinTrig trig, uint32_t* label, int32_t* seed_size, int32_t* queue, hipStream_t s) {
    (void)h;
    if (seed_cap == 0) return 0;
    hipLaunchKernelGGL(flood_ordered_kernel, dim3(1), dim3(64), 0, s, dx, dy, dmask, w, seed_idx, seed_bin, seed_thr,
                       d_n_seeds, seed_cap, trig, label, seed_size, queue);
    LR_HIP(hipGetLastError());
    return 0;
}

namespace {

// =============================================================================================
// Mode 1: parallel rounds, exact.
//
// The ordered semantics above are a well-founded recursion on the seed index: the flood of
// seed k is the connected set around it in {response_k > thr_k} minus everything claimed by
// seeds < k.  Rounds evaluate that recursion for many seeds at once:
//
//   explore  every active seed below the round's window (a prefix of the seed order; usually all of them) walks
//            its WHOLE reachable set w.r.t. the labels committed so far (its "footprint", a superset of its final
//            flood), records it per 8x8 tile in a private table and, at the end of the walk, stamps each pixel
//            with atomicMin(label, MARK|k).  A seed that finds a lower stamp, or whose stamp is overwritten by a
//            lower seed, is "blocked": a lower active seed can reach one of its pixels, so its flood is not
//            decided yet.  Footprints are never truncated at other seeds' stamps (a truncated walk would hide an
//            overlap from a third seed).
//   decide   unblocked seeds have footprints disjoint from every lower active footprint, so
//            their footprint IS their final flood: commit.  If some seed could not finish its
//            walk (private storage exhausted) only seeds below the lowest such seed commit.
//   commit   stamps of committed seeds become labels, all other stamps are erased; seeds whose
//            own pixel got committed by someone else are dead (the reference skips them).
//
// The lowest active seed is never blocked, so every round commits at least one seed; on real
// frames 4-8 rounds finish everything (longest chain of overlapping footprints with increasing
// index).  If a round makes no progress (only possible when storage is exhausted) the host
// finishes the remaining seeds with the ordered kernel, which is always exact.
// =============================================================================================

constexpr uint32_t kMarkBit = 0x80000000u;
// Per-wave LDS storage of a walk, two tiers: every seed starts with a 128-record frontier ring and a 256-tile table
// (5.4 KB of LDS, 90 VGPRs: 20 walks per CU); the few walks that outgrow it (edges over ~1500 px) start again in a workgroup with a
// 1024-record ring and a 2048-tile table (41 KB: 3 per CU) before the global slabs are the last resort.
#ifndef LR_RING_T
#define LR_RING_T 128
#endif
// (first-tier table: 256 tiles.  With 128 -- 5.8 KB of LDS per walk instead of 9.5, 20 walks per CU instead of 17 -- too
// many walks outgrow the tier: 6.6-6.7 Gpix/s instead of 7.5-7.6 on the bench, single frames 4.4 ms instead of 3.2.)
#ifndef LR_HASH_T
#define LR_HASH_T 256
#endif
constexpr int kRingT = LR_RING_T, kHashT = LR_HASH_T;
constexpr int kRingBig = 1024, kHashBig = 2048;
constexpr uint32_t kHandTiles = kHashT * 3 / 4, kHandRecs = kRingT;  // what a first-tier walk can hold when it is handed over
constexpr uint32_t kHandTable = 4, kHandRing = kHandTable + 3 * kHandTiles;
static_assert(kHandRing + 3 * kHandRecs <= kFloodHandWords, "hand-over record");
// Way-points (round 4).  A walk of many tiles is a long dependent chain -- a line's frontier is two records wide, so not even
// a team of wavefronts gets through it faster than a tile at a time -- and a blocked seed walks its footprint again every
// round: rounds 2 to 5 of the bench frame lasted 235, 181, 157 us for single walks of 183, 155, 142 tiles while the chip
// idled.  So a finished walk of at least wp_min_tiles tiles leaves kWpK pixels spread over its footprint (every
// ntiles / (kWpK + 1)-th tile in order of insertion, which for a line alternates between its two arms), and the seed's
// next walk starts from the seed AND from those pixels at once, on the eight wavefronts of a team: team_walk<.., true>.
constexpr uint32_t kWpK = (uint32_t)kFloodWpWords - 1u;
constexpr uint32_t kWpNone = 0xFFFFFFFFu;
constexpr uint32_t kWpThinPx = 40u;      // way-points only for footprints of at most this many pixels a tile (lines, not regions)
constexpr uint32_t kWpMaxTiles = 600u;   // the team's table holds an entry per (tile, source) and one per tile: 1536 in all
constexpr uint32_t kSrcShift = 28u, kSrcMask = 0xF0000000u, kSrcClaim = 0xF0000000u;  // table keys of a multi-source walk
constexpr uint32_t kBigCap = 8192;  // seeds per round that can move to the second tier (FloodBuffers::big_list)
constexpr uint32_t kFlagIncomplete = 1u, kFlagSelfFail = 2u;
constexpr uint32_t kHugeFlood = 1u << 14;  // (kernels_fit.hip: kSortLdsBig)
constexpr uint32_t kFullTiles = 48u;     // a first-tier walk is held at this many tiles once the round's second-tier list is full (explore_seed)
constexpr uint32_t kGiantProbes = 512u;  // marked seeds a round lets walk again (one team each: a launch's worth)
constexpr uint32_t kLogShrunk = 0x80000000u;  // FloodArgs::log_len: the log has been cut down to a later footprint by flood_rewalk_kernel
constexpr uint32_t kMaxSteps = 1u << 22;  // safety net of the walk loop: more records than an 8K frame has tile visits

struct FloodArgs {
    const float* dx;
    const float* dy;
    const uint8_t* dmask;
    int w, h, tiles_x;
    const int32_t* seed_idx;
    const int32_t* seed_bin;
    const float* seed_thr;
    uint32_t* label;
    uint32_t* blocked;  // per seed
    uint32_t* count;    // per seed: pixels walked this round
    uint32_t* flags;    // per seed
    uint8_t* tier;      // per seed, kept across rounds: 1 = go to the second storage tier at once
    uint32_t* ctrl;     // kCtrl* words
    uint8_t* dirty;     // per 256 consecutive pixels of the label image: stamped in this round (see the commit pass)
    uint4* slab_ring;   // n_slabs x slab_ring_cap records {tile, -, mask.lo, mask.hi}
    uint4* slab_hash;   // n_slabs x slab_hash_cap x 2 records {generation, tile+1, V.lo, V.hi} {A.lo, A.hi, -, -}
    uint32_t n_slabs, slab_ring_cap, slab_hash_cap;  // caps are powers of two
    uint32_t win_shift;                              // staged start (see kCtrlWindow): growth of the window per round
    uint32_t from_end;                               // explore the active list from its end (see flood_explore_kernel)
    uint32_t no_rest;                                // this round has no `rest` launch: entries past the grid are not walked
    uint32_t next_reach;                             // list entries the NEXT round's exploration reaches (0xFFFFFFFF: all)
    uint32_t big_cap;                                // seeds per round the second tier takes (0: tier switched off)
    uint32_t g_cap;                                  // partial-commit walks stop after this many tile steps
    uint32_t t1_tiles;                               // first tier hands a walk to the second at this many tiles (when there is one)
    uint32_t* handover;                              // state of the walks handed to the second tier (FloodBuffers::handover)
    uint32_t team_tiles;                             // test hook: the team's table counts as full at this many tiles
    uint32_t* blk;                                   // per seed: the lower seed a long, log-less walk of it was blocked by (0xFFFFFFFF: none; see kCtrlDeferLow)
    uint32_t defer_steps;                            // ... walks of at least this many tile steps
    uint32_t giant_many;                             // walks held back after which a frame counts as one of overlapping giants (see kCtrlStaged)
    uint32_t hold_min_big;                           // walks in the second tier after which the hold-back engages
    uint32_t hold_release;                           // the hold-back ends when so few seeds below the line are still active
    uint32_t t1_regional, t1_regional_min;           // ... at t1_regional tiles once the frame has had t1_regional_min walks beyond the first tier's table
    uint32_t t1_wide_tiles, t1_wide_front;           // ... or at this many tiles when its frontier holds this many records
    uint32_t* waypoints;                             // FloodBuffers::waypoints (kFloodWpWords per seed), wp_cap seeds
    uint32_t wp_cap;
    uint32_t wp_min_tiles;                           // walks of at least this many tiles leave way-points (0xFFFFFFFF: never)
    uint32_t* multi_list;                            // way-point seeds of the coming round (written by the survivors pass), kBigCap entries
    uint32_t multi_next;                             // the coming round walks that list in a launch of its own, beside its exploration
    // ---- re-walks from the log (flood_rewalk_kernel): a finished walk of at least log_min_tiles tiles leaves its footprint
    // as (tile, walked pixels) records; the seed's later rounds work on those records instead of walking the image again
    uint32_t log_min_tiles;                          // 0xFFFFFFFF: no logs
    uint32_t log_walk_tiles;                         // a seed with a log walks this many tiles before it turns to the log
    uint32_t log_use;                                // 0: this round's walks leave logs but none is used yet (FloodBuffers::log_from_round)
    uint32_t log_max_len;                            // logs of at most this many records are written and used (what the launched kernels' tables hold)
    uint32_t log_sweep;                              // test hook: every footprint is worked out by sweeps (flood_rewalk_kernel)
    uint32_t* host_progress;                         // FloodBuffers::host_progress (nullptr: nobody is looking)
    uint32_t* host_ctrl;                             // FloodBuffers::host_ctrl (with host_progress only)
    uint32_t quiet;                                  // this round goes without the second tier's launch on a frame that has one (flood_enqueue: calm hint)
    uint32_t giant_hold;                             // 1: only the lowest active seed walks on into a slab (see kCtrlLowest)
    uint32_t log_seeds;                              // seeds with per-seed words below (FloodBuffers::log_seeds)
    uint32_t* log_off;                               // first record of the seed's log ...
    uint32_t* log_len;                               // ... and their number (0: none)
    uint32_t* log_buf;                               // records: tile, walked lo, walked hi
    uint32_t log_cap;                                // records the buffer holds
    // ---- the giant step (see kCtrlGiantStep): the lowest active seed's flood by the whole device
    uint32_t giant_step;                             // 1: a walk of the LOWEST active seed that outgrows the LDS tiers asks for it instead of moving into a slab
    uint32_t* act_a;                                 // the two active lists: a round with work reads the one its index (kCtrlRounds) selects and
    uint32_t* act_b;                                 // appends to the other (rounds enqueued behind a request for a giant step do nothing and do not count)
    unsigned long long* giant_mask;                  // per 8x8 tile: the seed's acceptable pixels, then its flood's (FloodBuffers::giant_mask)
    uint32_t* giant_parent;                          // per pixel: union-find parent of the in-tile components' first pixels (FloodBuffers::giant_parent)
};
// All words but kCtrlGen are set up by flood_init_seeds_kernel every frame; kCtrlGen lives on for the lifetime of
// the slab memory (hash entries are tagged with it, so a generation must never be reused while old entries are
// around).  The rounds schedule themselves from these words: the host enqueues a batch of rounds blindly and
// looks at the block once per batch.
enum {
    kCtrlBarrier = 0,  // lowest seed whose walk ran out of storage this round
    kCtrlSlabs = 1,    // slabs handed out this round
    kCtrlNAct = 2,     // length of the current active list (0: nothing left to do)
    kCtrlNCommit = 3,  // seeds committed this round
    kCtrlNNext = 4,    // length of the next active list (being appended)
    kCtrlPhase = 5,    // hold-back of the weakest seeds: 0 not engaged, 1 holding, 2 released (see flood_advance_kernel)
    kCtrlDone = 6,     // workgroups of the survivors pass that have finished (the last one closes the round)
    kCtrlRounds = 7,   // rounds that had work
    kCtrlGen = 8,
    kCtrlStall = 9,    // a round made no progress: kCtrlNRemain seeds are left for the ordered tail
    kCtrlNRemain = 10,
    // Staged start: a round only walks seeds below this index (all lower active seeds are then walked too, so a
    // commit is as valid as in a full round); it grows by << win_shift per round up to the seed count.  Seeds are
    // ordered by strength, and most weak seeds sit on an edge that a strong seed's flood takes: they die without
    // ever having walked it.
    kCtrlWindow = 11,
    kCtrlNBig = 12,  // seeds handed to the second storage tier this round
    kCtrlBigTotal = 13,   // ... over the frame (diagnostics: lr_stage_counters)
    kCtrlSlabTotal = 14,  // slabs handed out over the frame
    kCtrlBelow = 15,      // active seeds below the window after this round (the window opens fully when none is left)
    // The host never learns the seed count before the frame's single synchronisation: everything that depends on it
    // is worked out by flood_init_seeds_kernel and kept here.
    kCtrlNSeeds = 16,   // seeds of this frame (clamped to the capacity the seed sort ran with)
    kCtrlWinHold = 17,  // the window stops here while the weakest seeds are held back
    kCtrlWalked = 18,   // [18..19] 64-bit: pixels walked by all explorations of the frame (diagnostics: re-walk factor)
    kCtrlSteps = 20,    // [20..21] 64-bit: tile steps of all explorations
    kCtrlBigSeen = 22,  // 1: the frame had many long walks (kCtrlBigLong) when the current round began -- early hand-over
    kCtrlBarrierNext = 24,  // lowest seed of the next round's list that its exploration launch will not reach (no `rest` launch)
    kCtrlBigLong = 23,  // walks of the frame that really outgrew the first tier (more tiles than its table holds)
    kCtrlMulti = 25,    // re-walks that started from several way-points at once (diagnostics: lr_stage_counters [10])
    kCtrlNMulti = 26,   // length of the current round's list of way-point seeds (walked by a launch of their own on a second stream)
    kCtrlNMultiNext = 27,  // ... of the next round's (being appended by the survivors pass)
    kCtrlLogTotal = 28,  // footprint-log records handed out this frame
    kCtrlLogWalks = 29,  // re-walks from logs (diagnostics: lr_stage_counters [11])
    kCtrlLogGiveUp = 30, // ... that gave their log up (tables too small) [12]
    // Giants (round 4): a walk that outgrows even the second tier's table (1536 tiles: a region of ~100 000 pixels) used to
    // move into a global slab at once -- and on a frame of smooth ramps hundreds of weak seeds share one such region, each
    // walking all of it in every round (a 4K ramp frame: 390 M pixels walked for 3.4 M labelled, 288 ms).  Now only the
    // LOWEST active seed may go on into a slab; any other is marked (tier bit 2), counts as a walk that did not finish, and
    // the window closes in front of the lowest marked seed until everything below it is resolved -- then it is the lowest
    // active seed, walks alone and commits, and takes the other marked seeds of its region with it unwalked.  Any prefix
    // of the seed order is a valid window, so this is the hold-back with a line the frame finds for itself.
    kCtrlLowest = 31,        // lowest active seed of the current round's list
    kCtrlLowestNext = 32,    // ... of the next round's (survivors pass)
    kCtrlGiantLow = 33,      // lowest active marked seed (0xFFFFFFFF: none)
    kCtrlGiantLowNext = 34,
    kCtrlGiantRuled = 35,    // 1: the current round's window was cut by this rule
    kCtrlGiants = 36,        // walks held back this way over the frame (diagnostics: lr_stage_counters [13])
    kCtrlGiantCountNext = 37,  // marked seeds among the survivors (survivors pass)
    // The giant step (round 5).  The lowest active seed is never blocked and always commits: its flood is a plain connected
    // component of a static predicate (acceptable for this seed and not committed), the one flood the WHOLE device can work on
    // without touching exactness.  When the lowest active seed is a marked giant -- at the end of a round (flood_advance), or
    // after a giant step (giant_finish_kernel) -- this word holds seed + 1: every kernel of a round leaves at once while it is
    // set (rounds enqueued blindly behind the request do nothing and do not count), the host sees the request in
    // report (flood_report) and enqueues the step (giant_*_kernel: tile masks, union-find over the tiles' components, labels).
    kCtrlGiantStep = 38,
    kCtrlGiantDone = 39,       // giant steps of the frame (diagnostics: lr_stage_counters [14]; the host waits for its count)
    kCtrlGiantPx = 40,         // pixels of the step in progress
    kCtrlGiantBlocks = 41,     // workgroups of giant_finish_kernel that have finished
    kCtrlMaxFlood = 43,        // pixels of the largest flood committed so far, if more than kHugeFlood (the fit's launches for huge components: a bit of flood_report)
    // A frame that holds back giant_many walks in a round with the full window goes on like a staged start (FloodBuffers::
    // win_first_shift): the next round's window is the strongest quarter of the seeds, and it doubles from round to round.
    kCtrlStaged = 44,
    kCtrlDeferLow = 45,        // lowest survivor that waits for its blocker (FloodArgs::blk): the window ends in front of it
    kCtrlDeferLowNext = 46,    // ... of the next round (survivors pass)
    kCtrlWindowFree = 42,      // the coming round's window before the giants' rule cut it (giant_finish_kernel applies the rule again)
    kCtrlQuietMiss = 47,       // walks that outgrew the first tier in a round enqueued without the second (FloodArgs::quiet)
    kCtrlTeamDone = 48,        // second-tier walks of this frame's first round that are over, held ones included ...
    kCtrlTeamGiants = 49,      // ... and how many of them were held back as giants (giants_all)
    kCtrlGiantReuse = 52,      // the giant step asked for can use the tile masks and the union-find of the step before it (same test, nothing changed but that step's commit)
    kCtrlWords = 56,
};
static_assert(kCtrlWords == kFloodCtrlWords, "control block size");

// Frontier records, table entries and ballots are the same in all 64 lanes.  Saying so (readfirstlane) lets the
// compiler keep them in SGPRs and run the bit-board logic on the scalar unit instead of the vector ALU.
__device__ __forceinline__ uint32_t uni(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ __forceinline__ uint64_t uni64(uint32_t lo, uint32_t hi) { return ((uint64_t)uni(hi) << 32) | uni(lo); }
// Lane masks straight from the comparison (v_cmp writes the SGPR pair), and a lane mask used as a per-lane condition.
// __ballot(a && b) goes through a 0/1 register and a second compare per ballot; the walk's conditions are combined as
// masks on the scalar unit instead.  (All 64 lanes are active wherever these are used.)
__device__ __forceinline__ uint64_t m_eq(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 32); }
__device__ __forceinline__ uint64_t m_ne(uint32_t a, uint32_t b) { return __builtin_amdgcn_uicmp(a, b, 33); }
__device__ __forceinline__ uint64_t m_ne64(uint64_t a, uint64_t b) { return __builtin_amdgcn_uicmpl(a, b, 33); }
__device__ __forceinline__ uint64_t m_lt_s(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 40); }
__device__ __forceinline__ uint64_t m_ge_s(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 39); }
__device__ __forceinline__ uint64_t m_gt_s(int a, int b) { return __builtin_amdgcn_sicmp(a, b, 38); }
__device__ __forceinline__ uint64_t m_gt_f(float a, float b) { return __builtin_amdgcn_fcmpf(a, b, 2); }
__device__ __forceinline__ bool lane_of(uint64_t mask) { return __builtin_amdgcn_inverse_ballot_w64(mask); }
__device__ __forceinline__ uint64_t uni64(uint64_t v) { return uni64((uint32_t)v, (uint32_t)(v >> 32)); }

// kCtrlStaged: the frame's FIRST round is in progress with a window beyond the strongest quarter, and giant_many walks have been held back
__device__ __forceinline__ bool giants_many(const uint32_t* ctrl, uint32_t giant_many) {
    const uint32_t n_seeds = uni(ctrl[kCtrlNSeeds]);
    return giant_many != 0u && uni(ctrl[kCtrlStaged]) == 0u && uni(ctrl[kCtrlRounds]) == 0u && n_seeds >= 8192u &&
           uni(ctrl[kCtrlWindow]) > (n_seeds >> 2) && uni(ld_agent(&ctrl[kCtrlGiants])) >= giant_many;
}
// The frame's first round, and nearly every second-tier walk that has ended so far ended held back (seven in eight, sixty-four at
// least): a frame that is ALL giants -- a noiseless gradient whose every pixel is a seed of one magnitude: 8192 walks that each
// fill the team's table before they give up, 7.9 ms of a 13 ms frame -- and the walks not yet begun are not begun
// (flood_explore_team_kernel); the giant steps that follow label the floods whatever their seeds found out.
__device__ __forceinline__ bool giants_all(const uint32_t* ctrl) {
    if (uni(ctrl[kCtrlRounds]) != 0u) return false;
    const uint32_t g = uni(ld_agent(&ctrl[kCtrlTeamGiants])), t = uni(ld_agent(&ctrl[kCtrlTeamDone]));
    return g >= 64u && g * 8u >= t * 7u;
}
// the active list a round with work reads, and the one it appends to (see FloodArgs::act_a)
__device__ __forceinline__ const uint32_t* act_now(const FloodArgs& A) { return (uni(A.ctrl[kCtrlRounds]) & 1u) ? A.act_b : A.act_a; }
__device__ __forceinline__ uint32_t* act_other(const FloodArgs& A) { return (uni(A.ctrl[kCtrlRounds]) & 1u) ? A.act_a : A.act_b; }
// a giant step has been asked for: the round's kernels leave at once
__device__ __forceinline__ bool giant_pending(const FloodArgs& A) { return uni(A.ctrl[kCtrlGiantStep]) != 0u; }

struct WalkState {
    uint32_t head, tail, cnt, ntiles;
    bool blocked;
    uint32_t steps;  // frontier records popped (diagnostics)
    uint32_t fresh;  // partial-commit walk: pixels it turned from stamps into labels
    uint32_t blocker = 0xFFFFFFFFu;  // lowest seed whose stamp the walk's own stamps met (the second tier's walks: FloodArgs::blk)
};

// The walk works on 8x8 pixel tiles, one tile per step, one pixel per lane: the acceptance test of the
// whole tile is a 64-bit ballot, connectivity inside the tile is a handful of scalar shift/and
// operations on that mask, and the frontier holds (tile, entry-mask) records instead of pixels.
// Each wave keeps a private table tile -> (V = pixels it has walked, A = pixels that pass the acceptance
// test): a record that re-enters a known tile is resolved from the table alone, without touching memory,
// and walked pixels stamped by lower seeds are remembered there too, so the walk terminates.
// All store operations below are wave-uniform (every lane performs the same access).
template <int kRing, int kHash, typename OrdT>
struct LdsStoreT {
    static constexpr bool kDeferStamps = true;
    static constexpr int kRingN = kRing, kHashN = kHash;
    static constexpr int kHashShift = 32 - __builtin_ctz((unsigned)kHash);
    uint32_t* rt;   // ring: tile
    uint32_t* rlo;  // ring: entry mask
    uint32_t* rhi;
    uint32_t* hk;   // table: tile + 1 (0 = empty)
    uint32_t* hv0;  // V
    uint32_t* hv1;
    OrdT* ord;      // table slots in order of insertion (what stamp_footprint walks)
    __device__ void note_new(uint32_t i, uint32_t slot) { ord[i] = (OrdT)slot; }
    __device__ uint32_t ring_cap() const { return kRing; }
    __device__ uint32_t hash_limit() const { return kHash * 3 / 4; }
    __device__ void get(uint32_t i, uint32_t& tile, uint64_t& m) const {
        const uint32_t j = i & (kRing - 1);
        tile = uni(rt[j]);
        m = uni64(rlo[j], rhi[j]);
    }
    __device__ void put(uint32_t i, uint32_t tile, uint64_t m) {
        const uint32_t j = i & (kRing - 1);
        rt[j] = tile;
        rlo[j] = (uint32_t)m;
        rhi[j] = (uint32_t)(m >> 32);
    }
    // returns true if the tile is known; slot = where it is or where it would go
    __device__ bool lookup(uint32_t tile, uint32_t& slot, uint64_t& V) const {
        const uint32_t key = tile + 1u;
        uint32_t hs = (key * 2654435761u) >> kHashShift;
        uint32_t cur = uni(hk[hs]);
        if (cur != key && cur != 0u) {  // a collision: rare, and kept out of the straight path
            for (int probe = 1; probe < kHash && cur != key && cur != 0u; ++probe) {
                hs = (hs + 1) & (kHash - 1);
                cur = uni(hk[hs]);
            }
        }
        slot = hs;
        V = 0ull;
        if (cur != key) return false;  // (the table is never full: hash_limit)
        V = uni64(hv0[hs], hv1[hs]);
        return true;
    }
    __device__ void value(uint32_t slot, uint64_t& V) const { V = uni64(hv0[slot], hv1[slot]); }
    __device__ void update(uint32_t slot, uint32_t tile, uint64_t V) {
        hk[slot] = tile + 1u;
        hv0[slot] = (uint32_t)V;
        hv1[slot] = (uint32_t)(V >> 32);
    }
};

using LdsStore = LdsStoreT<kRingT, kHashT, uint8_t>;
using LdsStoreBig = LdsStoreT<kRingBig, kHashBig, uint16_t>;

struct SlabStore {
    static constexpr bool kDeferStamps = false;
    __device__ void note_new(uint32_t, uint32_t) {}
    uint4* ring;
    uint4* hash;
    uint32_t rcap, hcap, gen;
    __device__ uint32_t ring_cap() const { return rcap; }
    __device__ uint32_t hash_limit() const { return hcap / 4 * 3; }
    __device__ static uint4 ld(const uint4* p) {
        uint4 v;
        const uint32_t* q = reinterpret_cast<const uint32_t*>(p);
        v.x = __hip_atomic_load(q + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.y = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.z = __hip_atomic_load(q + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        v.w = __hip_atomic_load(q + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return v;
    }
    __device__ static void st(uint4* p, uint4 v) {
        uint32_t* q = reinterpret_cast<uint32_t*>(p);
        __hip_atomic_store(q + 0, v.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 1, v.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 2, v.z, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(q + 3, v.w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __device__ void get(uint32_t i, uint32_t& tile, uint64_t& m) const {
        const uint4 v = ld(&ring[i & (rcap - 1)]);
        tile = uni(v.x);
        m = uni64(v.z, v.w);
    }
    __device__ void put(uint32_t i, uint32_t tile, uint64_t m) {
        st(&ring[i & (rcap - 1)], make_uint4(tile, 0u, (uint32_t)m, (uint32_t)(m >> 32)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
    // (records keep their stride of two 16-byte words; the second one is unused since the table holds V only)
    __device__ bool lookup(uint32_t tile, uint32_t& slot, uint64_t& V) const {
        const uint32_t key = tile + 1u;
        uint32_t hs = (key * 2654435761u) & (hcap - 1);
        for (uint32_t probe = 0; probe < hcap; ++probe) {
            const uint4 v = ld(&hash[2 * hs]);
            if (uni(v.x) != gen) break;  // another generation's record reads as empty
            if (uni(v.y) == key) {
                slot = hs;
                V = uni64(v.z, v.w);
                return true;
            }
            hs = (hs + 1) & (hcap - 1);
        }
        slot = hs;
        V = 0ull;
        return false;
    }
    __device__ void value(uint32_t slot, uint64_t& V) const {
        const uint4 v = ld(&hash[2 * slot]);
        V = uni64(v.z, v.w);
    }
    __device__ void update(uint32_t slot, uint32_t tile, uint64_t V) {
        st(&hash[2 * slot], make_uint4(gen, tile + 1u, (uint32_t)V, (uint32_t)(V >> 32)));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    }
};

// Frontier de-duplication: a small direct-mapped table (tile -> ring index of its not-yet-fetched record).
// A thin ridge re-enters the same tile from several neighbours; merging those entries into the pending
// record saves a whole step (a memory round trip) per duplicate.
constexpr int kPend = 64;
struct Pending {
    uint32_t* pt;  // tile + 1
    uint32_t* pi;  // ring index
    template <class Store>
    __device__ __forceinline__ void push(Store& S, WalkState& st, uint32_t tile, uint64_t m) {
        const uint32_t hs = ((tile + 1u) * 2654435761u) >> 26;  // 6 bits
        const uint32_t idx = uni(pi[hs]);
        // mergeable iff that record is still in the frontier (ring index in [head, tail))
        if (uni(pt[hs]) == tile + 1u && (int32_t)(idx - st.head) >= 0 && (int32_t)(st.tail - idx) > 0) {
            uint32_t t2;
            uint64_t m2;
            S.get(idx, t2, m2);
            if (t2 == tile) {
                S.put(idx, tile, m2 | m);
                return;
            }
        }
        S.put(st.tail, tile, m);
        pt[hs] = tile + 1u;
        pi[hs] = st.tail;
        st.tail += 1;
    }
};

// bit y of b -> bit 8*y (a byte's bits spread down a bit-board column)
__device__ __forceinline__ uint64_t spread_col(uint64_t b) {
    b = (b | (b << 28)) & 0x0000000F0000000Full;
    b = (b | (b << 14)) & 0x0003000300030003ull;
    b = (b | (b << 7)) & 0x0101010101010101ull;
    return b;
}

// Per-lane constants of push8 (direction d = lane & 7: 0 up, 1 down, 2 left, 3 right, 4 up-left, 5 up-right,
// 6 down-left, 7 down-right), computed once per walk.
struct PushLane {
    uint32_t off;    // neighbour tile id - tile id
    uint32_t shamt;  // where the direction's bits sit in the ring-hit ballot
    uint32_t msk;    // 0xFF for an edge, 1 for a corner
    uint32_t sh;     // shift of the (spread) bits into the neighbour's bit-board
    bool spread;     // left/right edges: the byte runs down a column
    bool dl;         // lane carries a direction
};
__device__ __forceinline__ PushLane push_lane(int lane) {
    const int d = lane & 7;
    const int dy = (d < 2) ? (d == 0 ? -1 : 1) : (d < 4 ? 0 : (d < 6 ? -1 : 1));
    const int dx = (d < 2) ? 0 : ((d & 1) ? 1 : -1);
    PushLane c;
    c.off = (uint32_t)(dy * 0x10000 + dx);
    c.shamt = (d < 4) ? 8u * (uint32_t)d : 32u + (uint32_t)(d - 4);
    c.msk = (d < 4) ? 0xFFu : 1u;
    c.sh = (uint32_t)((0x0007383F00070038ull >> (8 * d)) & 0xFFull);  // 56 0 7 0 63 56 7 0
    c.spread = d == 2 || d == 3;
    c.dl = lane < 8;
    // keep them in registers: recomputing the selects inside the walk loop costs more than four VGPRs
    asm volatile("" : "+v"(c.off), "+v"(c.shamt), "+v"(c.msk), "+v"(c.sh));
    return c;
}

// The first record a step appended, handed to the next step in registers when the frontier was empty before
// (a walk along a line: the chain tile -> neighbour tile then never waits for the ring or the table).
struct Forward {
    bool valid;
    uint32_t tile, slot;
    uint64_t entry;
    bool known;
};

// All (up to eight) neighbour records of a step at once, one direction per lane 0..7 (LDS store only): the
// per-direction entry masks are cut out of the ring-hit ballot H with lane-parallel arithmetic; the tile table
// (is the neighbour known, and what of it is walked) and the pending table are read by the eight lanes in one
// LDS round trip, and new records are appended with one ballot.
template <int kRing, int kHash, typename OrdT>
__device__ __forceinline__ void push8(LdsStoreT<kRing, kHash, OrdT>& S, Pending& P, WalkState& st, uint32_t tile,
                                      uint64_t H, int lane, const PushLane& c, Forward& fw) {
    const uint32_t nt = tile + c.off;
    const uint32_t key = nt + 1u;
    const uint64_t src = (H >> c.shamt) & (uint64_t)c.msk;
    uint64_t E = (c.spread ? spread_col(src) : src) << c.sh;
    uint32_t ts = (key * 2654435761u) >> LdsStoreT<kRing, kHash, OrdT>::kHashShift;
    const uint32_t hs = (key * 2654435761u) >> 26;
    // one round trip: first probe of the tile table with its walked set, and the pending entry
    uint32_t hk0 = S.hk[ts];
    uint32_t v0 = S.hv0[ts], v1 = S.hv1[ts];
    const uint32_t pt = P.pt[hs], pi = P.pi[hs];
    // lane masks from here on (lanes 0..7 carry a direction): what happens to a direction is decided on the scalar unit,
    // and a whole branch is skipped when no lane takes it
    const uint64_t m_want = m_ne64(E, 0ull) & 0xFFull;
    uint64_t m_found = m_eq(hk0, key);
    uint64_t m_search = m_want & ~m_found & m_ne(hk0, 0u);
    for (int probe = 1; probe < kHash && m_search != 0ull; ++probe) {  // collisions: rare
        if (lane_of(m_search)) {
            ts = (ts + 1) & (kHash - 1);
            hk0 = S.hk[ts];
        }
        const uint64_t hit = m_search & m_eq(hk0, key);
        if (lane_of(hit)) {
            v0 = S.hv0[ts];
            v1 = S.hv1[ts];
        }
        m_found |= hit;
        m_search &= ~hit & m_ne(hk0, 0u);
    }
    // A neighbour the wave already knows needs a record only for entry pixels it has not walked yet: every walked
    // set V is a union of whole in-tile components, so entries inside V could add nothing.  This drops the record
    // back into the tile a step came from (about half of all records otherwise).
    if (lane_of(m_found)) E &= ~(((uint64_t)v1 << 32) | v0);
    const uint64_t m_active = m_want & m_ne64(E, 0ull);
    // A record of the same tile that is still in the frontier (ring index in [head, tail)) takes the entries: the
    // pending entry was written with that record, and a ring slot is not reused while its index is in the window.
    const uint64_t m_merge = m_active & m_eq(pt, key) & m_ge_s((int32_t)(pi - st.head), 0) & m_gt_s((int32_t)(st.tail - pi), 0);
    if (m_merge != 0ull) {
        if (lane_of(m_merge)) {
            const uint32_t j = pi & (kRing - 1);
            atomicOr(&S.rlo[j], (uint32_t)E);
            atomicOr(&S.rhi[j], (uint32_t)(E >> 32));
        }
    }
    const uint64_t mf = m_active & ~m_merge;  // directions that get a record of their own
    if (mf != 0ull) {
        if (lane_of(mf)) {
            const uint32_t pos = st.tail + __builtin_amdgcn_mbcnt_hi((uint32_t)(mf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf, 0u));
            const uint32_t j = pos & (kRing - 1);
            S.rt[j] = nt;
            S.rlo[j] = (uint32_t)E;
            S.rhi[j] = (uint32_t)(E >> 32);
            P.pt[hs] = key;
            P.pi[hs] = pos;
        }
        const int f = __builtin_ctzll(mf);  // the lane whose record went to index st.tail
        fw.valid = true;
        fw.tile = (uint32_t)__builtin_amdgcn_readlane((int)nt, f);
        fw.slot = (uint32_t)__builtin_amdgcn_readlane((int)ts, f);
        fw.entry = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(E >> 32), f) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)E, f);
        fw.known = (m_found >> f) & 1ull;
    }
    st.tail += (uint32_t)__popcll(mf);
}

// scalar form (global slab store)
__device__ __forceinline__ void push8(SlabStore& S, Pending& P, WalkState& st, uint32_t tile, uint64_t H, int lane,
                                      const PushLane& c, Forward& fw) {
    (void)lane;
    (void)c;
    (void)fw;
    const uint64_t up = H & 0xFFull, dn = (H >> 8) & 0xFFull, lf = (H >> 16) & 0xFFull, rt = (H >> 24) & 0xFFull;
    auto send = [&](uint32_t nt, uint64_t E) {  // same filter as the LDS form: skip entries the neighbour has walked
        uint32_t slot;
        uint64_t V;
        if (S.lookup(nt, slot, V)) E &= ~V;
        if (E != 0ull) P.push(S, st, nt, E);
    };
    if (up) send(tile - 0x10000u, up << 56);          // (x,-1) -> pixel (x,7) of the tile above
    if (dn) send(tile + 0x10000u, dn);                // (x,8)  -> pixel (x,0) of the tile below
    if (lf) send(tile - 1u, spread_col(lf) << 7);     // (-1,y) -> pixel (7,y) of the left tile
    if (rt) send(tile + 1u, spread_col(rt));          // (8,y)  -> pixel (0,y) of the right tile
    if ((H >> 32) & 1ull) send(tile - 0x10001u, 1ull << 63);
    if ((H >> 33) & 1ull) send(tile - 0xFFFFu, 1ull << 56);
    if ((H >> 34) & 1ull) send(tile + 0xFFFFu, 1ull << 7);
    if ((H >> 35) & 1ull) send(tile + 0x10001u, 1ull);
}

// The 36 pixels around an 8x8 tile, one per lane 0..35: lanes 0-7 the row above (x = 0..7), 8-15 the row below,
// 16-23 the column to the left (y = 0..7), 24-31 the column to the right, 32-35 the corners (-1,-1) (8,-1)
// (-1,8) (8,8).  A record is pushed to a neighbour tile only for ring pixels that are themselves acceptable and
// touch a newly walked pixel, so no step is spent on a tile that nothing can enter.
__device__ __forceinline__ void ring_xy(int lane, int& rx, int& ry) {
    const int g = lane >> 3, i = lane & 7;
    rx = (g == 0 || g == 1) ? i : (g == 2 ? -1 : (g == 3 ? 8 : ((i & 1) ? 8 : -1)));
    ry = (g == 0) ? -1 : (g == 1 ? 8 : ((g == 2 || g == 3) ? i : ((i & 2) ? 8 : -1)));
}
// in-tile pixels (bit = y*8+x) 8-adjacent to this lane's ring pixel
__device__ __forceinline__ uint64_t ring_adjacency(int lane) {
    if (lane >= 36) return 0ull;
    int rx, ry;
    ring_xy(lane, rx, ry);
    uint64_t m = 0ull;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = rx + dx, y = ry + dy;
            if (x >= 0 && x < 8 && y >= 0 && y < 8) m |= 1ull << (y * 8 + x);
        }
    return m;
}
// pixels of the tile (bit = y*8+x) that are 8-adjacent to this lane's pixel, and the pixel itself
__device__ __forceinline__ uint64_t tile_neighbours(int lane) {
    const int lr = lane >> 3, lc = lane & 7;
    uint64_t m = 0ull;
    for (int dy = -1; dy <= 1; ++dy)
        for (int dx = -1; dx <= 1; ++dx) {
            const int x = lc + dx, y = lr + dy;
            if (x >= 0 && x < 8 && y >= 0 && y < 8) m |= 1ull << (y * 8 + x);
        }
    return m;
}
// What one lane holds of a frontier record before it is processed.  For a tile the wave already knows, the
// table answers (V, A, ring) and no memory is touched; otherwise the lane's pixel of the tile and its pixel of
// the surrounding ring are loaded.
struct TileFetch {
    uint32_t tile, slot;
    uint64_t entry, V;
    size_t q;
    float dx, dy, rdx, rdy;
    uint32_t dm, rdm;
    uint32_t lab, rlab;  // label words of the lane's pixel / ring pixel (own-aware walks and the partial-commit walk only)
    bool known;
    uint64_t inside, rinside;  // lane masks: the lane's pixel of the tile / of the ring lies in the frame
};

// `fw` (when valid) is the record at ring index i as the previous step's push8 left it in registers.
// element `idx` of a uniform array through a 32-bit byte offset (global_load with scalar base + vector offset)
__device__ __forceinline__ uint32_t ld32(const uint32_t* p, uint32_t idx) {
    return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(p) + (idx << 2));
}
__device__ __forceinline__ float ldf(const float* p, uint32_t idx) {
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(p) + (idx << 2));
}
__device__ __forceinline__ uint32_t ld8(const uint8_t* p, uint32_t idx) { return p[idx]; }

// per-lane constants of the address computation: offset of the lane's pixel from the tile's first pixel, and of its
// ring pixel (may be negative: wraps modulo 2^32 and comes out right after the add)
struct LaneGeom {
    uint32_t off, roff;
};

// kMode 0: exploration (direction mask and gradients; the label words too if the seed already owns pixels, `own`);
// kMode 1: the partial-commit walk, which looks at nothing but the label words.
template <int kMode, class Store>
__device__ __forceinline__ TileFetch fetch_tile(const FloodArgs& A, const Store& S, uint32_t i, int lr, int lc,
                                                int rx, int ry, bool ring_lane, const Forward& fw,
                                                const LaneGeom& G, bool own) {
    TileFetch f;
    if (fw.valid) {
        f.tile = fw.tile;
        f.entry = fw.entry;
    } else {
        S.get(i, f.tile, f.entry);
    }
    // Addresses: the tile's first pixel is a wave-uniform index (scalar unit), the lane's place in the tile and in
    // the ring around it are per-lane constants of the walk (G.off, G.roff): one 32-bit add per pixel, and the loads
    // take a uniform base with a 32-bit byte offset (frames have fewer than 2^29 pixels: launch_filter checks).
    const int ty = (int)(f.tile >> 16), tx = (int)(f.tile & 0xFFFFu);  // tile id = ty << 16 | tx
    const uint32_t base = (uint32_t)(ty * 8) * (uint32_t)A.w + (uint32_t)(tx * 8);
    f.inside = m_lt_s(ty * 8 + lr, A.h) & m_lt_s(tx * 8 + lc, A.w);
    const uint32_t q = lane_of(f.inside) ? base + G.off : 0u;
    f.q = q;
    // The pixels are requested whether or not the wave has been in this tile before (one step in fourteen is a
    // revisit): the table then holds the walked pixels only -- 12 bytes a tile instead of 28, which is what lets more
    // walks share a CU -- and the loads are on their way before the table is looked at.
    // (no label load: the commit pass clears the direction mask of every pixel it commits, so "not claimed by an
    // earlier flood" is part of the mask test)
    f.dm = 0u;
    f.dx = f.dy = 0.f;
    f.lab = kLabelFree;
    if constexpr (kMode == 0) {
        f.dm = ld8(A.dmask, q);
        f.dx = ldf(A.dx, q);
        f.dy = ldf(A.dy, q);
    }
    if (kMode == 1 || own) f.lab = ld32(A.label, q);  // (wave-uniform condition)
    const int rr = ty * 8 + ry, rc = tx * 8 + rx;
    // (one unsigned comparison per coordinate: a negative one reads as a huge number; lanes 36..63 have no ring pixel)
    f.rinside = __builtin_amdgcn_uicmp((uint32_t)rr, (uint32_t)A.h, 36) & __builtin_amdgcn_uicmp((uint32_t)rc, (uint32_t)A.w, 36) &
                0xFFFFFFFFFull;
    const uint32_t rq = lane_of(f.rinside) ? base + G.roff : 0u;
    f.rdm = 0u;
    f.rdx = f.rdy = 0.f;
    f.rlab = kLabelFree;
    if constexpr (kMode == 0) {
        f.rdm = ld8(A.dmask, rq);
        f.rdx = ldf(A.dx, rq);
        f.rdy = ldf(A.dy, rq);
    }
    if (kMode == 1 || own) f.rlab = ld32(A.label, rq);
    if (fw.valid) {
        f.slot = fw.slot;
        f.known = fw.known;
        f.V = 0ull;
        if (f.known) S.value(f.slot, f.V);
    } else {
        f.known = S.lookup(f.tile, f.slot, f.V);
    }
    return f;
}

// Walks the footprint of seed k from the state in `st`.  Returns 0 when the walk is complete, 1 when
// the store ran out; `st` then holds a resumable state.
//
// A step is: acceptance ballots of the tile (from the loads issued when its record was popped), connected
// closure of the entry pixels, table update, records for the neighbours, pop of the next record and issue of
// its loads.  Stamping is not part of the dependent chain tile -> neighbour tile: which seeds meet on a pixel
// does not depend on when the stamps land within a round, so the LDS store only remembers the walked pixels (V)
// and stamps them all at the end (stamp_footprint), with many atomics in flight at once.  The slab store
// stamps as it goes.
#ifdef LR_WALK_TIMING
// Diagnostic build (LR_EXTRA_FLAGS=-DLR_WALK_TIMING, printed by LIBRECTIFY_FLOOD_DEBUG): where a step of a long walk
// spends its time.  Sums over walks of more than 100 steps: [0] walks, [1] steps, then s_memtime ticks (about one per
// cycle; every reading costs some 150 itself) [2] waiting for the tile's pixels, [3] closure, [4] table update,
// [5] push, [6] pop, lookup and issue of the next loads; [7] steps that found their tile in the table.
__device__ unsigned long long g_walk_timing[8];
#define LR_TICK(i)                                        \
    {                                                     \
        const uint64_t t_ = __builtin_amdgcn_s_memtime(); \
        tacc[i] += t_ - tlast;                            \
        tlast = t_;                                       \
    }
#else
#define LR_TICK(i)
#endif

// kMode 0 explores (acceptance = direction mask and response; a seed that already owns pixels -- `own`, see
// flood_partial_commit_kernel -- also passes through the pixels labelled with its own index).  kMode 1 is the
// partial-commit walk: it accepts the pixels that carry this seed's stamp or its label, nothing else, and turns the
// stamps it reaches into labels; `dmask_rw` is the direction mask it clears there.
template <class Store, int kMode = 0>
__device__ int walk(const FloodArgs& A, uint32_t k, int b, float thr, float sn, float cs, Store& S, Pending& P,
                    WalkState& st, int lane, bool own = false, uint8_t* dmask_rw = nullptr, uint32_t tile_cap = 0xFFFFFFFFu,
                    uint32_t wide_tiles = 0xFFFFFFFFu, uint32_t wide_front = 0xFFFFFFFFu) {
    const uint32_t mine = kMarkBit | k;
    const int lr = lane >> 3, lc = lane & 7;
    int rx, ry;
    ring_xy(lane, rx, ry);
    const bool ring_lane = lane < 36;
    const uint64_t adj = ring_adjacency(lane);
    uint64_t nbr = tile_neighbours(lane);
    asm volatile("" : "+v"(nbr));  // a per-lane constant: keep it in registers
    if (st.head == st.tail) return 0;
    const uint32_t tile_limit = min(S.hash_limit(), tile_cap);
    if ((st.tail - st.head) + 8u > S.ring_cap() || st.ntiles + 2u > tile_limit) return 1;
    const PushLane pc = push_lane(lane);
    const uint32_t bin_bit = 1u << b;
    Forward fw;
    fw.valid = false;
    LaneGeom G;
    G.off = (uint32_t)(lr * A.w + lc);
    G.roff = (uint32_t)(ry * A.w + rx);
    asm volatile("" : "+v"(G.off), "+v"(G.roff));
    TileFetch cur = fetch_tile<kMode>(A, S, st.head, lr, lc, rx, ry, ring_lane, fw, G, own);
#ifdef LR_WALK_TIMING
    uint64_t tacc[5] = {0, 0, 0, 0, 0}, tlast = __builtin_amdgcn_s_memtime();
    uint32_t nknown = 0;
#endif
    for (;;) {
        st.head += 1;
        st.steps += 1;
#ifdef LR_WALK_TIMING
        nknown += cur.known ? 1u : 0u;
#endif
        const uint32_t tile = cur.tile;
        // acceptable pixels of the tile, and of the 36-pixel ring around it (lane order, see ring_xy)
        uint64_t Am, Rg;
        if constexpr (kMode == 0) {
            Am = cur.inside & m_ne(cur.dm & bin_bit, 0u) & m_gt_f(directional(cur.dx, cur.dy, sn, cs), thr);
            Rg = cur.rinside & m_ne(cur.rdm & bin_bit, 0u) & m_gt_f(directional(cur.rdx, cur.rdy, sn, cs), thr);
            if (own) {  // (wave-uniform) pixels this seed has committed already are its own ground
                Am |= cur.inside & m_eq(cur.lab, k);
                Rg |= cur.rinside & m_eq(cur.rlab, k);
            }
        } else {
            Am = cur.inside & (m_eq(cur.lab, mine) | m_eq(cur.lab, k));
            Rg = cur.rinside & (m_eq(cur.rlab, mine) | m_eq(cur.rlab, k));
        }
        LR_TICK(0)
        uint64_t R = cur.entry & Am;
        uint64_t New = 0ull;
        if (R != 0ull) {
            // Connected closure of the entry pixels inside the tile.  One iteration is "8-neighbour dilation of R, restricted to Am", evaluated
            // with a pixel per lane (an acceptable pixel joins when its 3x3 neighbourhood meets R): three vector
            // instructions instead of sixteen on the scalar unit, which the rest of the step keeps busy.
            const uint64_t reach = lane_of(Am) ? nbr : 0ull;
            for (;;) {  // two dilations per convergence test: the test (compare -> scalar compare -> branch) is the slow part
                const uint64_t R1 = m_ne64(R & reach, 0ull);
                R = m_ne64(R1 & reach, 0ull);
                if (R == R1) break;
            }
            New = R & ~cur.V;
        }
        LR_TICK(1)
        if (New != 0ull || !cur.known) {
            S.update(cur.slot, tile, cur.V | New);
            if (!cur.known) {
                S.note_new(st.ntiles, cur.slot);
                st.ntiles += 1;
            }
        }
        st.cnt += (uint32_t)__popcll(New);
        if constexpr (kMode == 1) {
            // stamps of this seed reached from its seed pixel through its own stamps and labels only: no lower active seed
            // can reach them (it would have stamped them), so they belong to this seed's flood whatever happens elsewhere
            const uint64_t fresh = New & m_eq(cur.lab, mine);
            if (lane_of(fresh)) {
                A.label[cur.q] = k;
                dmask_rw[cur.q] = 0;
            }
            st.fresh += (uint32_t)__popcll(fresh);
        }
        LR_TICK(2)
        fw.valid = false;
        const bool was_empty = st.head == st.tail;  // then the first record appended now is the next one popped
        if (New != 0ull) {
            if constexpr (!Store::kDeferStamps && kMode == 0) {
                uint32_t old = kLabelFree;
                if (lane_of(New)) {
                    old = atomicMin(&A.label[cur.q], mine);
                    A.dirty[cur.q >> 8] = 1;
                }
                bool foreign = false;
                if (old > mine) {  // free, or stamped by a higher seed that is hereby blocked
                    if (old != kLabelFree) A.blocked[old & ~kMarkBit] = 1u;
                } else if (old < mine && old >= kMarkBit) {  // a lower active seed reaches this pixel too
                    foreign = true;
                }
                if (__ballot(foreign)) st.blocked = true;
            }
            // ring pixels that are acceptable and touch a newly walked pixel become entries of their own tiles
            const uint64_t H = Rg & m_ne64(New & adj, 0ull);
            if (H != 0ull) push8(S, P, st, tile, H, lane, pc, fw);
        }
        fw.valid = fw.valid && was_empty;
        LR_TICK(3)
#ifdef LR_WALK_TIMING
        if (st.head == st.tail && st.steps > 100 && lane == 0) {
            atomicAdd(&g_walk_timing[0], 1ull);
            atomicAdd(&g_walk_timing[1], (unsigned long long)st.steps);
            for (int i = 0; i < 5; ++i) atomicAdd(&g_walk_timing[2 + i], (unsigned long long)tacc[i]);
            atomicAdd(&g_walk_timing[7], (unsigned long long)nknown);
        }
#endif
        if (st.head == st.tail) return 0;
        if ((st.tail - st.head) + 8u > S.ring_cap() || st.ntiles + 2u > tile_limit) return 1;
        // a walk with a wide frontier (a region, not a line) is handed to the second tier early: its team of wavefronts
        // takes a frontier eight records at a time
        if (st.ntiles >= wide_tiles && (st.tail - st.head) >= wide_front) return 1;
        if (st.steps > kMaxSteps) return 1;  // never reached by a terminating walk; treated like exhausted storage
        if (kMode == 1 && st.steps >= A.g_cap) return 1;  // partial-commit walk: any connected part is as safe as the whole
        cur = fetch_tile<kMode>(A, S, st.head, lr, lc, rx, ry, ring_lane, fw, G, own);
        LR_TICK(4)
    }
}

// Stamps every pixel the LDS walk has covered: atomicMin(label, MARK|k) on each, eight tiles' worth in flight
// before the first result is looked at.  A lower stamp found means a lower active seed reaches the pixel (this
// seed is blocked); a higher stamp replaced means that seed is blocked.
template <class Lds>
__device__ __forceinline__ void stamp_footprint(const FloodArgs& A, uint32_t k, const Lds& S, WalkState& st,
                                                int lane, uint32_t first = 0u, uint32_t stride = 8u) {
    const uint32_t mine = kMarkBit | k;
    const int lr = lane >> 3, lc = lane & 7;
    bool foreign = false;
    uint32_t fmin = 0xFFFFFFFFu;
    // (a team of wavefronts shares the tiles: each takes eight at `first`, `first + stride`, ...)
    for (uint32_t i0 = first; i0 < st.ntiles; i0 += stride) {
        uint32_t old[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            old[j] = kLabelFree;
            const uint32_t i = i0 + (uint32_t)j;
            if (i < st.ntiles) {
                const uint32_t slot = S.ord[i];
                const uint32_t tile = S.hk[slot] - 1u;
                const uint64_t V = ((uint64_t)S.hv1[slot] << 32) | S.hv0[slot];
                if ((V >> lane) & 1ull) {
                    const size_t q = (size_t)((tile >> 16) * 8 + lr) * A.w + ((tile & 0xFFFFu) * 8 + lc);
                    old[j] = atomicMin(&A.label[q], mine);
                    A.dirty[q >> 8] = 1;
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (old[j] > mine) {
                if (old[j] != kLabelFree) A.blocked[old[j] & ~kMarkBit] = 1u;
            } else if (old[j] < mine && old[j] >= kMarkBit) {
                foreign = true;
                fmin = min(fmin, old[j] & ~kMarkBit);
            }
        }
    }
    if (__ballot(foreign)) {
        st.blocked = true;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) fmin = min(fmin, (uint32_t)__shfl_xor((int)fmin, off));
        st.blocker = min(st.blocker, fmin);
    }
}

// Way-points of a finished walk (see kWpK): lanes 0 .. kWpK-1 take the (lane + 1) ntiles / (kWpK + 1)-th tile in order of
// insertion and a walked pixel of it.  Written once per seed: footprints only shrink, the points stay representative.
template <class Lds>
__device__ __forceinline__ void save_waypoints(const FloodArgs& A, uint32_t k, const Lds& S, uint32_t ntiles, int lane) {
    uint32_t* wp = A.waypoints + (size_t)k * kFloodWpWords;
    if ((uint32_t)lane < kWpK) {
        const uint32_t i = ((uint32_t)lane + 1u) * ntiles / (kWpK + 1u);
        const uint32_t slot = S.ord[i];
        const uint32_t tile = S.hk[slot] - 1u;
        const uint64_t V = ((uint64_t)S.hv1[slot] << 32) | S.hv0[slot];
        uint32_t q = kWpNone;  // (a tile whose entry pixels turned out unacceptable holds nothing)
        if (V != 0ull) {
            const uint32_t bit = (uint32_t)__builtin_ctzll(V);
            q = ((tile >> 16) * 8u + (bit >> 3)) * (uint32_t)A.w + (tile & 0xFFFFu) * 8u + (bit & 7u);
        }
        wp[1 + lane] = q;
    }
    if (lane == 0) wp[0] = kWpK | (ntiles << 8);
}

// The records of a finished walk (see FloodArgs::log_buf): handed out of one buffer per frame by a counter.  A buffer that
// is full leaves the seed without a log: it walks.
template <class Lds>
__device__ __forceinline__ void save_log(const FloodArgs& A, uint32_t k, const Lds& S, uint32_t ntiles, int lane) {
    uint32_t off = 0u;
    if (lane == 0) off = atomicAdd(&A.ctrl[kCtrlLogTotal], ntiles);
    off = (uint32_t)__shfl((int)off, 0);
    if (off + ntiles > A.log_cap || off + ntiles < off) return;
    uint32_t* rec = A.log_buf + (size_t)off * 3u;
    for (uint32_t i = (uint32_t)lane; i < ntiles; i += 64u) {
        const uint32_t slot = S.ord[i];
        rec[3u * i] = S.hk[slot] - 1u;
        rec[3u * i + 1u] = S.hv0[slot];
        rec[3u * i + 2u] = S.hv1[slot];
    }
    if (lane == 0) {
        A.log_off[k] = off;
        A.log_len[k] = ntiles;
    }
}

// One seed's exploration by one wavefront (see walk).  kFirstTier: a walk that outgrows the store is handed to the
// second tier (big_list) instead of going on in a slab.
template <class Lds, bool kFirstTier>
__device__ __forceinline__ void explore_seed(const FloodArgs& A, const BinTrig& trig, uint32_t k, Lds& L, Pending& P,
                                             uint32_t* __restrict__ big_list, int lane, uint32_t t1_tiles = 0xFFFFFFFFu) {
    const int s = (int)uni((uint32_t)A.seed_idx[k]);
    const int b = (int)uni((uint32_t)A.seed_bin[k]);
    const float thr = __uint_as_float(uni(__float_as_uint(A.seed_thr[k])));
    const float sn = trig.st[b], cs = trig.ct[b];
    // the four values at the seed pixel in one round trip (most seeds of a late round end right here)
    const uint32_t seed_label = A.label[s];
    const uint32_t seed_mask = A.dmask[s];
    const float seed_dx = A.dx[s], seed_dy = A.dy[s];
    // A seed whose pixel carries its OWN index has committed a part of its flood in an earlier round (partial commit,
    // flood_partial_commit_kernel) and explores on from there: its walk passes through the pixels labelled with its index.
    const bool own = seed_label == k;
    if (seed_label < kMarkBit && !own) return;  // claimed by an earlier flood: dead (the survivors pass takes it off the list)
    // on this round's list of way-point seeds: a team walks it right now, in a launch beside this one (enqueue_round)
    // (bit 3: finished by a giant step between the rounds -- the survivors pass of this round takes it off the list)
    if (kFirstTier && (uni((uint32_t)A.tier[k]) & 10u) != 0u) return;
    if (!own && !(((seed_mask >> b) & 1) && directional(seed_dx, seed_dy, sn, cs) > thr)) {
        if (lane == 0) A.flags[k] = kFlagSelfFail;  // flood() accepts nothing, not even the seed
        return;
    }
    WalkState st{0u, 1u, 0u, 0u, false, 0u, 0u};
    int rc = 1;
    // a walk that outgrew the first tier in an earlier round does not try it again (footprints only shrink, but
    // rarely below 190 tiles from above 1500 px)
    // ... and a seed that left way-points on a long footprint in an earlier round goes straight to the second tier, whose
    // team walks it from all of them at once
    // (not on a frame that went on staged after its first round -- kCtrlStaged: what a weak seed reached then says little
    // about what it reaches once the stronger seeds have committed)
    const bool outgrown = kFirstTier && A.big_cap != 0u && (uni((uint32_t)A.tier[k]) & 1u) != 0u && uni(A.ctrl[kCtrlStaged]) == 0u;
    const bool wp_seed = kFirstTier && A.big_cap != 0u && A.wp_min_tiles != 0xFFFFFFFFu && k < A.wp_cap &&
                         uni(A.waypoints[(size_t)k * kFloodWpWords]) != 0u;
    // A seed with a log (save_log) walks a few tiles only: most footprints have shrunk to a handful of tiles by their second
    // round, and a short walk is cheaper than the records of a long one.  If the walk is not over by then, the seed goes
    // on this round's list of flood_rewalk_kernel, which runs behind the exploration (nothing is stamped yet).
    const uint32_t log_w = (kFirstTier && A.log_min_tiles != 0xFFFFFFFFu && k < A.log_seeds) ? uni(A.log_len[k]) : 0u;
    const uint32_t log_n = log_w & ~kLogShrunk;
    const bool has_log = log_n != 0u && log_n <= A.log_max_len && A.log_use != 0u;
    const bool skip_first = (outgrown || wp_seed) && !has_log;
    if (!skip_first) {
        for (int i = lane; i < Lds::kHashN; i += 64) L.hk[i] = 0u;
        P.pt[lane] = 0u;
        const int sr = s / A.w, sc = s - sr * A.w;
        L.put(0u, ((uint32_t)(sr >> 3) << 16) | (uint32_t)(sc >> 3), 1ull << ((sr & 7) * 8 + (sc & 7)));
        // (with a second tier behind it, the first hands a walk over at A.t1_tiles tiles, before its table is full: the
        // second tier's team of wavefronts is the faster walker from there on)
        const bool hand_over = kFirstTier && A.big_cap != 0u;
        if (has_log) {
            // (a log that the last round has already cut down to the footprint of then, and that is still twice the budget
            // long, goes on the list at once: the walk would only find out the same, 12 steps later)
            const bool direct = (log_w & kLogShrunk) != 0u && log_n >= 2u * A.log_walk_tiles;
            if (!direct) rc = walk(A, k, b, thr, sn, cs, L, P, st, lane, own, nullptr, A.log_walk_tiles);
            if (rc != 0) {
                uint32_t pos = 0;
                if (lane == 0) pos = atomicAdd(&A.ctrl[kCtrlNMulti], 1u);
                pos = (uint32_t)__shfl((int)pos, 0);
                if (pos < kBigCap) {
                    if (lane == 0) A.multi_list[pos] = k;
                    return;
                }
            }
        }
        if (rc != 0)  // (from the start, or from where the budgeted walk stands: the list was full)
            rc = walk(A, k, b, thr, sn, cs, L, P, st, lane, own, nullptr, hand_over ? t1_tiles : 0xFFFFFFFFu,
                      hand_over ? A.t1_wide_tiles : 0xFFFFFFFFu, hand_over ? A.t1_wide_front : 0xFFFFFFFFu);
    }
    if (kFirstTier && rc != 0 && A.big_cap == 0u && A.quiet != 0u) {
        // outgrew the first tier in a round that was enqueued without the second (the last frame never needed it): nothing is
        // stamped yet, the seed counts as unfinished -- nothing above it commits this round -- and the report says "not calm",
        // so the rounds that follow bring the second tier
        if (lane == 0) {
            A.flags[k] = kFlagIncomplete;
            atomicMin(&A.ctrl[kCtrlBarrier], k);
            atomicAdd(&A.ctrl[kCtrlQuietMiss], 1u);
        }
        return;
    }
    if (kFirstTier && rc != 0 && A.big_cap != 0u) {
        // outgrew the first tier: start again in the second (nothing is stamped yet, so nothing to undo)
        uint32_t pos = 0;
        if (lane == 0) pos = atomicAdd(&A.ctrl[kCtrlNBig], 1u);
        pos = (uint32_t)__shfl((int)pos, 0);
        if (pos < A.big_cap) {
            if (lane == 0) {
                big_list[pos] = k;
                if (!skip_first || outgrown) {  // (a way-point seed has not outgrown anything: it does not count towards the hold-back)
                    A.tier[k] = 1;  // (bit 1 is clear here: listed seeds left above)
                    atomicAdd(&A.ctrl[kCtrlBigTotal], 1u);
                }
            }
            // The second tier goes on from where this walk stands (nothing is stamped yet): the walked sets of its tiles
            // and the frontier records travel with the list entry.  A seed that skipped this tier has nothing to hand over.
            uint32_t* hb = A.handover + (size_t)pos * kFloodHandWords;
            const uint32_t nt = skip_first ? 0u : st.ntiles, nr = skip_first ? 0u : st.tail - st.head;
            if (lane == 0) {
                hb[0] = nt;
                hb[1] = nr;
                hb[2] = st.steps;
            }
            for (uint32_t i = (uint32_t)lane; i < nt; i += 64u) {
                const uint32_t slot = L.ord[i];
                hb[kHandTable + 3u * i] = L.hk[slot];
                hb[kHandTable + 3u * i + 1u] = L.hv0[slot];
                hb[kHandTable + 3u * i + 2u] = L.hv1[slot];
            }
            for (uint32_t i = (uint32_t)lane; i < nr; i += 64u) {
                const uint32_t j = (st.head + i) & (uint32_t)(Lds::kRingN - 1);
                hb[kHandRing + 3u * i] = L.rt[j];
                hb[kHandRing + 3u * i + 1u] = L.rlo[j];
                hb[kHandRing + 3u * i + 2u] = L.rhi[j];
            }
            return;
        }
        if (skip_first) {  // no room in the second tier this round: walk in the first after all, then a slab
            for (int i = lane; i < Lds::kHashN; i += 64) L.hk[i] = 0u;
            P.pt[lane] = 0u;
            const int sr = s / A.w, sc = s - sr * A.w;
            L.put(0u, ((uint32_t)(sr >> 3) << 16) | (uint32_t)(sc >> 3), 1ull << ((sr & 7) * 8 + (sc & 7)));
            rc = walk(A, k, b, thr, sn, cs, L, P, st, lane, own);
        }
        // second tier full this round: carry on in a slab from the state reached -- if this is the lowest active seed; any
        // other is held back like a walk that outgrows the second tier's table (see kCtrlLowest): with eight thousand walks in
        // the second tier the frame is one of overlapping giants (a noiseless radial gradient: 35 837 seeds, all of one
        // magnitude, sixteen rings), and the slabs are for the one walk that is sure to commit
        // (with the giant step the lowest active seed is marked as well: the end of the round finds the lowest survivor marked
        // and asks for the step -- flood_advance)
        if (rc != 0 && A.giant_hold != 0u && (A.giant_step != 0u || k != uni(A.ctrl[kCtrlLowest]))) {
            if (lane == 0) {
                A.tier[k] = (uint8_t)(A.tier[k] | 5u);  // (outgrew the first tier; a giant)
                A.flags[k] = kFlagIncomplete;
                atomicMin(&A.ctrl[kCtrlBarrier], k);
                atomicAdd(&A.ctrl[kCtrlGiants], 1u);
            }
            return;
        }
    }
    stamp_footprint(A, k, L, st, lane);
    // (only THIN footprints: a region's frontier is wide, the team's level-synchronous walk takes it eight tiles at a time as
    // it is, and the second table entry per tile only costs -- natural 4K frame: flood 1.45 -> 1.61 ms without this test)
    if (kFirstTier && rc == 0 && !wp_seed && st.ntiles >= A.wp_min_tiles && st.cnt <= kWpThinPx * st.ntiles && k < A.wp_cap)
        save_waypoints(A, k, L, st.ntiles, lane);
    // the footprint's records for the seed's later rounds (flood_rewalk_kernel); one log per seed and frame: the records of
    // ANY finished walk of the seed hold its present footprint
    if (kFirstTier && rc == 0 && st.ntiles >= A.log_min_tiles && st.ntiles <= A.log_max_len && k < A.log_seeds && uni(A.log_len[k]) == 0u)
        save_log(A, k, L, st.ntiles, lane);
    if (rc != 0) {
        // LDS storage exhausted: move the walk to a global slab and carry on
        uint32_t slab = 0;
        if (lane == 0) slab = atomicAdd(&A.ctrl[kCtrlSlabs], 1u);
        slab = (uint32_t)__shfl((int)slab, 0);
        if (slab < A.n_slabs) {
            uint32_t gen = 0;
            if (lane == 0) atomicAdd(&A.ctrl[kCtrlSlabTotal], 1u);
            if (lane == 0) gen = atomicAdd(&A.ctrl[kCtrlGen], 1u) + 1u;
            gen = (uint32_t)__shfl((int)gen, 0);
            SlabStore G{A.slab_ring + (size_t)slab * A.slab_ring_cap, A.slab_hash + (size_t)slab * A.slab_hash_cap * 2,
                        A.slab_ring_cap, A.slab_hash_cap, gen};
            for (uint32_t i = st.head; i != st.tail; ++i) {
                uint32_t t;
                uint64_t m;
                L.get(i, t, m);
                G.put(i, t, m);
            }
            for (int i = 0; i < Lds::kHashN; ++i) {
                const uint32_t key = L.hk[i];
                if (key) {
                    uint32_t slot;
                    uint64_t v0;
                    (void)G.lookup(key - 1u, slot, v0);
                    G.update(slot, key - 1u, ((uint64_t)L.hv1[i] << 32) | L.hv0[i]);
                }
            }
            rc = walk(A, k, b, thr, sn, cs, G, P, st, lane, own);
        }
        if (rc != 0 && lane == 0) {
            A.flags[k] = kFlagIncomplete;
            atomicMin(&A.ctrl[kCtrlBarrier], k);
        }
    }
    if (lane == 0) {
        A.count[k] = st.cnt;
        if (st.blocked) A.blocked[k] = 1u;
        A.flags[k] |= st.steps << 8;  // diagnostics only (LIBRECTIFY_FLOOD_DEBUG)
    }
}

// One wavefront per workgroup (walks differ in length by three orders of magnitude, and a workgroup keeps its
// LDS until its longest wave is done), one seed per wavefront.
//
// The host knows neither the seed count nor the length of the round's list when it enqueues a round blindly: the grid is
// its guess (the capacity of the seed sort; above kFullGridCap seeds halved from round to round as the lists shrink, with the
// `rest` launch behind every main launch for the entries past the guess), one entry per workgroup, so that the hardware's
// dispatcher balances the walks.  Rounds enqueued just in time know their list's length and take exactly that grid.
// (Measured and rejected: a chip-sized grid pulling entries through one atomic counter -- 40 000 same-address atomics
// serialise in L2, +0.4 ms per frame; one kernel whose workgroups stride over the list -- the loop around the walk
// costs 137 spilled registers, and the slowest workgroup's eight walks in a row make the first round half again as
// long (round 5 tried the loop again for the entries past the grid only: 96 -> 127 registers, four walks a SIMD instead of five,
// round one 321 -> 342 us, the batch 8 % slower); a grid of the full capacity in every round WHEN EIGHT OR NINE ROUNDS WERE
// ENQUEUED BLINDLY -- an empty workgroup costs the dispatcher 0.4 ns, 0.3 ms per frame that other frames' kernels wait for
// (with at most four blind rounds it is the rule now: enqueue_round); a smaller first storage tier (64-record ring, 128-tile table, 4.9 KB) for more
// walks in flight, handing longer walks to a second kernel -- the tiers' kernels run one after the other, so every
// round lasts as long as the longest walk of EACH tier: 1.43 -> 2.0 ms per flood.)
//
// Nine walks in ten belong to seeds that do not commit in their round (round one of the 4K test frame: 532 000 of
// 563 000 tile steps; LIBRECTIFY_FLOOD_DEBUG prints the split), and a blocked seed walks again every round.  "Parking"
// them was built and measured: a finished walk leaves its footprint as a (tile, mask) list, a blocked seed only
// re-stamps that list in later rounds (no dependent chain), and one that comes out unblocked on the list -- a superset
// of its present footprint -- commits by a walk of its own after the commit pass.  Exact (all tests passed), but the
// stale supersets block far more than footprints do, and seeds blocked by each other's stale stamps resolve one per
// round: 39-55 rounds instead of 6-9; with lists dropped as soon as one of their pixels is committed elsewhere, 9-15
// rounds and 2.3-3.4 ms instead of 1.5-2.6.  Not in the tree.
//
// What a step costs (LR_WALK_TIMING above; rocprofv3 counters in profiles/r02_pmc_flood_explore.txt): about 340
// instructions (168 VALU, 167 SALU, 16 LDS, 6 loads) and some 3000 cycles when the wave has its SIMD to itself, of which
// the wait for the tile's pixels is 250-400; closure 320, table update 50, push 850, pop + lookup + issue 850.  So a round
// is (longest walk) x (latency of the bookkeeping chain), not memory, and the bench frame's rounds show it: 373, 296,
// 231, 207, 139 us for longest walks of 174, 173, 155, 142, 97 steps.  Measured on that basis and NOT in the tree:
//  - tile-major planes ((dx, dy) interleaved, 8x8 tiles contiguous: 27 cache lines per step instead of 60).
//    tools/ubench/tile_gather.hip had promised 323 -> 143 us for round 1's gathers; the rounds came out at 373, 296,
//    231, 207, 139 us again, the batch throughput within noise (7.56-7.58 Gpix/s both, same box, alternating), and the
//    filter kernel went from 39.0 to 43.0 us (eight 64-byte pieces per store instead of one 512-byte run).
//  - requesting the pixels of the record behind the head one step early (right after this step's own have arrived, so
//    that the compiler's s_waitcnt vmcnt(0) does not wait for them too): 385, 304, 238, 207, 142 us.  The same with
//    throw-away loads that only warm the cache: 446, 370, 270, 240, 160 us.
template <bool kRest>
__device__ __forceinline__ void explore_body(const FloodArgs& A, const BinTrig& trig, uint32_t* __restrict__ big_list, uint32_t first) {
    __shared__ uint32_t s_ring[3][kRingT];
    __shared__ uint32_t s_hash[3][kHashT];
    __shared__ uint32_t s_pend[2][kPend];
    __shared__ uint8_t s_ord[kHashT];
#ifdef LR_LDS_PAD  // experiment: fewer walks per CU (what does occupancy buy?)
    __shared__ uint32_t s_pad[LR_LDS_PAD / 4];
    asm volatile("" ::"v"(&s_pad[threadIdx.x]) : "memory");
#endif
    const int lane = threadIdx.x & 63;
    if (giant_pending(A)) return;
    const uint32_t* __restrict__ act = act_now(A);
    const uint32_t n_act = uni(A.ctrl[kCtrlNAct]), window = uni(A.ctrl[kCtrlWindow]);
    // A frame with many LONG walks (sixteen beyond what the first tier's table holds: natural images, frames of long bars)
    // hands its walks over earlier from the next round on -- at 32 tiles instead of 190: the first tier's rounds last as
    // long as its longest walk, and the second tier has 512 teams to take the walks side by side (with 128 teams the long
    // bars' thousands of thin walks queued up behind each other and the same rule cost that frame 0.5 ms; with 512 it
    // gains 0.3).  The synthetic bench frames never get there (0-7 long walks), and handing THEIR walks over early costs
    // them 0.3-0.9 ms: the tiers' kernels run one after the other, and a thin walk gains nothing from a team.
    const uint32_t t1_usual = uni(A.ctrl[kCtrlBigSeen]) != 0u ? min(A.t1_tiles, A.t1_regional) : A.t1_tiles;
    // The second tier's list is full already when this workgroup starts -- eight thousand long walks in this round: a frame of
    // overlapping giants, a noiseless gradient whose every pixel is a seed -- so a walk that outgrows this tier will be held
    // back whatever its length (explore_seed): it is held at kFullTiles tiles instead of walking on to the table's 190 (radial
    // gradient at 1080p, 78 704 seeds: first round 45 -> 11 ms).  Read HERE, with the other words of the control block: the
    // same read in front of every walk cost the bench frames' first round 1.1 ms (0.3 -> 1.4).
    const bool tier2_full = A.giant_hold != 0u && A.big_cap != 0u && uni(A.ctrl[kCtrlNBig]) >= A.big_cap;
    const uint32_t lowest = uni(A.ctrl[kCtrlLowest]);
    LdsStore L{s_ring[0], s_ring[1], s_ring[2], s_hash[0], s_hash[1], s_hash[2], s_ord};
    Pending P{s_pend[0], s_pend[1]};
    // The list is walked from its end: the first round's list is in seed order, strongest first, and the longest
    // walks belong to the weak seeds at its end (low thresholds, large footprints).  Started first, they run
    // alongside the mass of short walks instead of after it.
    if (!kRest) {
        const uint32_t ai = uni(blockIdx.x);
        if (ai >= n_act) return;
        // (a round without a `rest` launch whose list is longer than its grid walks the FIRST entries: the survivors pass
        // that wrote the list has put a barrier at the lowest seed behind them -- enqueue_round)
        const bool fwd = A.no_rest != 0u && n_act > gridDim.x;
        const uint32_t k = uni(act[(A.from_end && !fwd) ? n_act - 1u - ai : ai]);
        if (k >= window) return;  // not yet in the staged window (stays active)
        const uint32_t t1_tiles = (tier2_full && k != lowest) ? min(t1_usual, kFullTiles) : t1_usual;
        explore_seed<LdsStore, true>(A, trig, k, L, P, big_list, lane, t1_tiles);
    } else {
        for (uint32_t ai = first + uni(blockIdx.x); ai < n_act; ai += gridDim.x) {
            const uint32_t k = uni(act[A.from_end ? n_act - 1u - ai : ai]);
            if (k >= window) continue;
            const uint32_t t1_tiles = (tier2_full && k != lowest) ? min(t1_usual, kFullTiles) : t1_usual;
            explore_seed<LdsStore, true>(A, trig, k, L, P, big_list, lane, t1_tiles);
        }
    }
}
// (90 VGPRs: five walks per SIMD, 20 per CU; the 5.4 KB of LDS would allow 30.  Capped at 80 or 72 registers (six or
// seven per SIMD) the compiler spills 9 or 18 of them and the rounds and the batch rate stay within 2 %: not taken.  With
// half the walks per CU -- LDS padded to 18.9 KB -- round one takes 600 us instead of 374 and the batch rate drops by a
// quarter, so occupancy is what the bulk rounds live on up to about this point.)
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void flood_explore_kernel(
    FloodArgs A, BinTrig trig, uint32_t* __restrict__ big_list) {
    explore_body<false>(A, trig, big_list, 0u);
}
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(4, 8))) void flood_explore_rest_kernel(
    FloodArgs A, BinTrig trig, uint32_t* __restrict__ big_list, uint32_t first) {
    explore_body<true>(A, trig, big_list, first);
}

// Second storage tier: the same walk from the start with a 1024-record ring and a 2048-tile table (dynamic LDS,
// kBigLdsBytes), for the seeds the first tier handed over this round.
constexpr size_t kBigLdsBytes = (size_t)(3 * kRingBig + 3 * kHashBig + 2 * kPend) * 4 + (size_t)kHashBig * 2;
__global__ __launch_bounds__(64) void flood_explore_big_kernel(FloodArgs A, BinTrig trig,
                                                               uint32_t* __restrict__ big_list) {
    extern __shared__ uint32_t s_big[];
    const int lane = threadIdx.x & 63;
    if (giant_pending(A)) return;
    const uint32_t ai = uni(blockIdx.x);
    const uint32_t n_big = uni(A.ctrl[kCtrlNBig]);
    if (ai >= (n_big < A.big_cap ? n_big : A.big_cap)) return;
    const uint32_t k = uni(big_list[ai]);
    uint32_t* ring = s_big;
    uint32_t* hash = ring + 3 * kRingBig;
    uint32_t* pend = hash + 3 * kHashBig;
    uint16_t* ord = reinterpret_cast<uint16_t*>(pend + 2 * kPend);
    LdsStoreBig L{ring, ring + kRingBig, ring + 2 * kRingBig, hash, hash + kHashBig, hash + 2 * kHashBig, ord};
    Pending P{pend, pend + kPend};
    explore_seed<LdsStoreBig, false>(A, trig, k, L, P, big_list, lane);
}

// ---- Second tier, cooperative: a TEAM of wavefronts walks one footprint -------------------------------------------------
// The walks that reach the second tier are the long ones (hundreds of tiles: the weak seeds of smooth regions and long
// edges), few per frame, and one wavefront takes them a tile at a time: on a natural 4K image the second-tier kernel was
// half of the flood (1.75 of 3.1 ms) for 52 walks.  Their frontiers are wide -- a footprint of 800 tiles is some 100
// tile-levels deep (tools/sim/flood_sim.cpp, SIM_PAR) -- so here the frontier is processed a LEVEL at a time by the
// eight wavefronts of a workgroup: records [begin, end) of the ring are this level, wavefront w takes records
// begin + w, begin + w + 8, ..., the records it pushes are appended behind `end` through an LDS counter and form the
// next level.  Two barriers per level.  The footprint (the fixed point) does not depend on the order of the steps:
//   - tile table: find-or-insert by compare-and-swap on the key; the walked set V of a tile only grows, by atomic OR;
//   - a step pushes neighbour records only for the pixels IT was the first to add to V (the OR returns the old bits): two
//     wavefronts that enter a tile at once share its new pixels between them instead of both walking on from all;
//   - a stale V read by a neighbour's filter only costs a record that finds nothing new.
// Stamps, pixel count and the blocked mark are taken from the table when the walk is over, by all wavefronts.  If ring or
// table run out, the team moves into a global slab -- table, unprocessed records and all -- and goes on there (TeamGlobalStore).
#ifndef LR_TEAM_WAVES
#define LR_TEAM_WAVES 8
#endif
constexpr int kTeamWaves = LR_TEAM_WAVES;
#ifndef LR_TEAM_RING
#define LR_TEAM_RING 1024
#endif
constexpr int kRingTeam = LR_TEAM_RING;
#ifndef LR_TEAM_GRID
#define LR_TEAM_GRID 512
#endif
constexpr uint32_t kTeamGrid = LR_TEAM_GRID;  // workgroups; each strides over the round's second-tier list
constexpr uint32_t kVoidTile = 0xFFFFFFFFu;

struct TeamShared {
    uint32_t tail, end, ntiles, blocked, overflow, cnt, steps, pad;
    uint32_t adj[8];          // multi-source walk: sources whose regions share a pixel with source i (bit per source)
    uint32_t reach, ctiles;   // ... sources connected to the seed's own; tiles of the footprint kept
    uint32_t blocker, pad2;   // lowest seed whose stamp this walk's stamps met (WalkState::blocker)
};
__device__ __forceinline__ uint32_t lds_now(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

struct TeamStore {
    static constexpr int kHashShift = 32 - __builtin_ctz((unsigned)kHashBig);
    uint32_t* rt;
    uint32_t* rlo;
    uint32_t* rhi;
    uint32_t* hk;
    uint32_t* hv0;
    uint32_t* hv1;
    uint16_t* ord;
    TeamShared* sh;
    __device__ uint32_t hash_limit() const { return kHashBig * 3 / 4; }
    __device__ void get(uint32_t i, uint32_t& tile, uint64_t& m) const {
        const uint32_t j = i & (kRingTeam - 1);
        tile = uni(rt[j]);
        m = uni64(rlo[j], rhi[j]);
    }
    // find or insert (wave-uniform; the atomics are lane 0's): true if the tile was known
    __device__ bool lookup(uint32_t tile, uint32_t& slot, uint64_t& V) const {
        const uint32_t key = tile + 1u;
        const bool first_lane = (threadIdx.x & 63u) == 0u;
        uint32_t hs = (key * 2654435761u) >> kHashShift;
        for (int probe = 0; probe < kHashBig; ++probe) {
            uint32_t cur = uni(lds_now(&hk[hs]));
            if (cur == 0u) {
                uint32_t old = 0u;
                if (first_lane) old = atomicCAS(&hk[hs], 0u, key);
                old = uni(old);
                if (old == 0u) {
                    uint32_t idx = 0u;
                    if (first_lane) idx = atomicAdd(&sh->ntiles, 1u);
                    idx = uni(idx);
                    if (first_lane) ord[idx] = (uint16_t)hs;
                    slot = hs;
                    V = 0ull;
                    return false;
                }
                cur = old;
            }
            if (cur == key) {
                slot = hs;
                V = uni64(lds_now(&hv0[hs]), lds_now(&hv1[hs]));
                return true;
            }
            hs = (hs + 1) & (kHashBig - 1);
        }
        slot = hs;  // (never: the table is kept below hash_limit)
        V = 0ull;
        return true;
    }
    __device__ void value(uint32_t slot, uint64_t& V) const { V = uni64(lds_now(&hv0[slot]), lds_now(&hv1[slot])); }
    // what team_walk / team_push8 need besides (the same names on the store in global memory below)
    __device__ uint32_t ring_cap() const { return (uint32_t)kRingTeam; }
    __device__ uint32_t hash_mask() const { return (uint32_t)(kHashBig - 1); }
    __device__ uint32_t hash_of(uint32_t key) const { return (key * 2654435761u) >> kHashShift; }
    __device__ uint32_t peek_key(uint32_t slot) const { return hk[slot]; }                              // (per lane)
    __device__ uint64_t peek_walked(uint32_t slot) const { return ((uint64_t)hv1[slot] << 32) | hv0[slot]; }  // (per lane)
    __device__ uint64_t add_walked(uint32_t slot, uint64_t New) const {  // one lane; returns the bits that were there
        uint32_t o0 = 0u, o1 = 0u;
        if ((uint32_t)New) o0 = atomicOr(&hv0[slot], (uint32_t)New);
        if ((uint32_t)(New >> 32)) o1 = atomicOr(&hv1[slot], (uint32_t)(New >> 32));
        return ((uint64_t)o1 << 32) | o0;
    }
    __device__ void write_record(uint32_t pos, uint32_t tile, uint64_t E) const {  // (per lane)
        const uint32_t j = pos & (uint32_t)(kRingTeam - 1);
        rt[j] = tile;
        rlo[j] = (uint32_t)E;
        rhi[j] = (uint32_t)(E >> 32);
    }
    __device__ void mark_processed(uint32_t i) const { rt[i & (uint32_t)(kRingTeam - 1)] = kVoidTile; }
    // read-only look-up, per lane (the walk is over: nothing is inserted any more)
    __device__ bool find_ro(uint32_t tile, uint32_t& slot) const {
        const uint32_t key = tile + 1u;
        uint32_t hs = (key * 2654435761u) >> kHashShift;
        for (int probe = 0; probe < kHashBig; ++probe) {
            const uint32_t cur = hk[hs];
            if (cur == key) {
                slot = hs;
                return true;
            }
            if (cur == 0u) return false;
            hs = (hs + 1) & (kHashBig - 1);
        }
        return false;
    }
};

// A multi-source walk keeps, in the SAME table, one entry per (tile, source) -- what that source has seen of the tile as
// connected to itself -- and one per tile for the pixels claimed so far by anybody; the source (0 = the seed itself,
// 1.. = way-points) or kSrcClaim sits in the four high bits of the tile id (tile rows stay below 4096: frames of fewer
// than 2^29 pixels).  Ring records carry the source the same way.  This view hands fetch_tile the plain tile id and
// directs its look-up to the entry of the record's source.
struct MultiView {
    const TeamStore& S;
    mutable uint32_t tag;
    __device__ void get(uint32_t i, uint32_t& tile, uint64_t& m) const {
        uint32_t t;
        S.get(i, t, m);
        tag = t & kSrcMask;
        tile = t & ~kSrcMask;
    }
    __device__ bool lookup(uint32_t tile, uint32_t& slot, uint64_t& V) const { return S.lookup(tile | tag, slot, V); }
    __device__ void value(uint32_t slot, uint64_t& V) const { S.value(slot, V); }
};

// The same store in GLOBAL memory (one overflow slab of FloodBuffers: 16 Ki ring records, a table of 64 Ki tiles), for a
// team whose walk has outgrown its LDS: frames without strong edges, where a single flood covers a smooth ramp of
// hundreds of thousands of pixels.  (One wavefront used to carry such a walk on alone, a tile at a time through global
// memory: 72 ms for a 2051x1153 frame of soft blobs.)  Only this workgroup touches the slab, its waves share one L1, and
// every access is an agent-scope atomic as in SlabStore; the counters stay in LDS.  The table is CLEARED by the team
// before use (1 MB of stores) instead of being tagged with a generation: find-or-insert is then a compare-and-swap on
// the key word alone, the walked set of a fresh entry is zero, and later users of the slab (generation-tagged) see
// nothing valid in it.  Entry layout as SlabStore's: uint4 (generation, key, walked lo, walked hi) at hash[2 * slot];
// the unused second uint4 of entry i holds the slot of the i-th tile inserted (what the stamping walks).
struct TeamGlobalStore {
    uint4* ring;
    uint4* hash;
    uint32_t rcap, hcap;
    TeamShared* sh;
    __device__ static uint32_t* w(uint4* p, int k) { return reinterpret_cast<uint32_t*>(p) + k; }
    __device__ static uint32_t ldw(uint4* p, int k) { return __hip_atomic_load(w(p, k), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ static void stw(uint4* p, int k, uint32_t v) { __hip_atomic_store(w(p, k), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    __device__ uint32_t hash_limit() const { return hcap / 4 * 3; }
    __device__ uint32_t ring_cap() const { return rcap; }
    __device__ uint32_t hash_mask() const { return hcap - 1u; }
    __device__ uint32_t hash_of(uint32_t key) const { return (key * 2654435761u) & (hcap - 1u); }
    __device__ void get(uint32_t i, uint32_t& tile, uint64_t& m) const {
        uint4* r = &ring[i & (rcap - 1u)];
        tile = uni(ldw(r, 0));
        m = uni64(ldw(r, 2), ldw(r, 3));
    }
    __device__ bool lookup(uint32_t tile, uint32_t& slot, uint64_t& V) const {
        const uint32_t key = tile + 1u;
        const bool first_lane = (threadIdx.x & 63u) == 0u;
        uint32_t hs = hash_of(key);
        for (uint32_t probe = 0; probe < hcap; ++probe) {
            uint4* e = &hash[2u * hs];
            uint32_t cur = uni(ldw(e, 1));
            if (cur == 0u) {
                uint32_t old = 0u;
                if (first_lane) old = atomicCAS(w(e, 1), 0u, key);
                old = uni(old);
                if (old == 0u) {
                    uint32_t idx = 0u;
                    if (first_lane) idx = atomicAdd(&sh->ntiles, 1u);
                    idx = uni(idx);
                    if (first_lane) stw(&hash[2u * idx + 1u], 0, hs);
                    slot = hs;
                    V = 0ull;
                    return false;
                }
                cur = old;
            }
            if (cur == key) {
                slot = hs;
                V = uni64(ldw(e, 2), ldw(e, 3));
                return true;
            }
            hs = (hs + 1u) & (hcap - 1u);
        }
        slot = hs;
        V = 0ull;
        return true;
    }
    __device__ void value(uint32_t slot, uint64_t& V) const { V = uni64(ldw(&hash[2u * slot], 2), ldw(&hash[2u * slot], 3)); }
    __device__ uint32_t peek_key(uint32_t slot) const { return ldw(&hash[2u * slot], 1); }
    __device__ uint64_t peek_walked(uint32_t slot) const { return ((uint64_t)ldw(&hash[2u * slot], 3) << 32) | ldw(&hash[2u * slot], 2); }
    __device__ uint64_t add_walked(uint32_t slot, uint64_t New) const {
        uint32_t o0 = 0u, o1 = 0u;
        if ((uint32_t)New) o0 = atomicOr(w(&hash[2u * slot], 2), (uint32_t)New);
        if ((uint32_t)(New >> 32)) o1 = atomicOr(w(&hash[2u * slot], 3), (uint32_t)(New >> 32));
        return ((uint64_t)o1 << 32) | o0;
    }
    __device__ void write_record(uint32_t pos, uint32_t tile, uint64_t E) const {
        uint4* r = &ring[pos & (rcap - 1u)];
        stw(r, 0, tile);
        stw(r, 2, (uint32_t)E);
        stw(r, 3, (uint32_t)(E >> 32));
    }
    __device__ void mark_processed(uint32_t i) const { stw(&ring[i & (rcap - 1u)], 0, kVoidTile); }
    __device__ uint32_t ord_slot(uint32_t idx) const { return ldw(&hash[2u * idx + 1u], 0); }
};

// neighbour records of a step, one direction per lane 0..7 (as push8; no merging with pending records of the same tile:
// duplicates of a level are taken by different wavefronts at the same time)
template <class Store>
__device__ __forceinline__ void team_push8(Store& S, uint32_t tile, uint64_t H, int lane, const PushLane& c, uint32_t tag = 0u) {
    const uint32_t nt = (tile + c.off) | tag;  // (tag: the source of a multi-source walk; lanes whose neighbour lies outside the frame carry no entry)
    const uint32_t key = nt + 1u;
    const uint64_t src = (H >> c.shamt) & (uint64_t)c.msk;
    uint64_t E = (c.spread ? spread_col(src) : src) << c.sh;
    uint32_t ts = S.hash_of(key);
    const uint64_t m_want = m_ne64(E, 0ull) & 0xFFull;
    uint32_t hk0 = 0u;
    uint64_t Vn = 0ull;
    if (lane_of(m_want)) hk0 = S.peek_key(ts);
    uint64_t m_found = m_want & m_eq(hk0, key);
    uint64_t m_search = m_want & ~m_found & m_ne(hk0, 0u);
    for (uint32_t probe = 1; probe <= S.hash_mask() && m_search != 0ull; ++probe) {  // collisions
        if (lane_of(m_search)) {
            ts = (ts + 1u) & S.hash_mask();
            hk0 = S.peek_key(ts);
        }
        const uint64_t hit = m_search & m_eq(hk0, key);
        m_found |= hit;
        m_search &= ~hit & m_ne(hk0, 0u);
    }
    if (lane_of(m_found)) Vn = S.peek_walked(ts);
    E &= ~Vn;  // entries the neighbour has walked add nothing
    const uint64_t mf = m_want & m_ne64(E, 0ull);
    if (mf != 0ull) {
        uint32_t base = 0u;
        if (lane == 0) base = atomicAdd(&S.sh->tail, (uint32_t)__popcll(mf));
        base = uni(base);
        if (lane_of(mf)) {
            const uint32_t pos = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(mf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mf, 0u));
            S.write_record(pos, nt, E);
        }
    }
}

// returns 0 when the footprint is complete; 1 when ring or table ran out: *begin_out is then the first ring index that
// may hold an unprocessed record (processed ones read kVoidTile)
// kMulti (TeamStore only): the records carry a source each -- the seed's own pixel and the way-points its last walk left
// (kWpK) -- and every source grows its region at once.  A source keeps what it has SEEN as connected to itself in entries of
// its own (V_s, whole in-tile components as ever) and goes on only from the pixels it was the first of all sources to CLAIM
// (the tile's claim entry, atomic OR): the regions share the footprint out between them and stop where they meet.  Two
// regions that touch have seen a common pixel: if p (claimed by s) and q (claimed by s') are neighbours, s pushed a record
// for q when it claimed p -- the neighbour filter looks at s's OWN entry of q's tile -- and q, though claimed already,
// entered V_s as well as V_s'.  After the walk (flood_explore_team_kernel) sources with a common pixel in some tile are
// joined, and the footprint is the union of V_s over the sources joined with source 0 -- exactly the pixels connected to
// the seed: each V_s is connected and contains its starting point; every pixel on a path from the seed is claimed by some
// source (induction along the path: a claimed pixel's in-tile component and ring neighbours are all visited by its
// claimer), and consecutive pixels of the path have claimers that saw a common pixel.  Way-points the footprint has lost
// grow regions of their own that join nothing and are dropped.
template <class Store, bool kMulti = false>
__device__ int team_walk(const FloodArgs& A, uint32_t k, int b, float thr, float sn, float cs, Store& S, int lane,
                         int wave, bool own, uint32_t first_level, uint32_t tile_cap, uint32_t* begin_out, uint32_t* steps_out) {
    TeamShared* sh = S.sh;
    const int lr = lane >> 3, lc = lane & 7;
    int rx, ry;
    ring_xy(lane, rx, ry);
    const bool ring_lane = lane < 36;
    const uint64_t adj = ring_adjacency(lane);
    uint64_t nbr = tile_neighbours(lane);
    asm volatile("" : "+v"(nbr));
    const PushLane pc = push_lane(lane);
    const uint32_t bin_bit = 1u << b;
    Forward fw;
    fw.valid = false;
    LaneGeom G;
    G.off = (uint32_t)(lr * A.w + lc);
    G.roff = (uint32_t)(ry * A.w + rx);
    asm volatile("" : "+v"(G.off), "+v"(G.roff));
    uint32_t gb = 0u, ge = first_level, steps = 0u, levels = 0u;  // the first level: the seed's record, or the frontier handed over
    int rc = 0;
    for (;;) {
        for (uint32_t i = gb + (uint32_t)wave; i < ge; i += kTeamWaves) {
            // room for what the steps in flight may add: eight records and one tile each
            if ((uni(lds_now(&sh->tail)) - gb) + 8u * kTeamWaves > S.ring_cap() ||
                uni(lds_now(&sh->ntiles)) + (kMulti ? 2u : 1u) * kTeamWaves + 1u > min(S.hash_limit(), tile_cap)) {
                if (lane == 0) sh->overflow = 1u;
                break;
            }
            if (uni(lds_now(&sh->overflow)) != 0u) break;
            uint32_t tag = 0u;
            TileFetch cur;
            if constexpr (kMulti) {
                const MultiView MV{S, 0u};
                cur = fetch_tile<0>(A, MV, i, lr, lc, rx, ry, ring_lane, fw, G, own);
                tag = MV.tag;
            } else {
                cur = fetch_tile<0>(A, S, i, lr, lc, rx, ry, ring_lane, fw, G, own);
            }
            ++steps;
            const uint32_t tile = cur.tile;
            uint64_t Am = cur.inside & m_ne(cur.dm & bin_bit, 0u) & m_gt_f(directional(cur.dx, cur.dy, sn, cs), thr);
            uint64_t Rg = cur.rinside & m_ne(cur.rdm & bin_bit, 0u) & m_gt_f(directional(cur.rdx, cur.rdy, sn, cs), thr);
            if (own) {
                Am |= cur.inside & m_eq(cur.lab, k);
                Rg |= cur.rinside & m_eq(cur.rlab, k);
            }
            uint64_t R = cur.entry & Am;
            if (R != 0ull) {
                const uint64_t reach = lane_of(Am) ? nbr : 0ull;
                for (;;) {
                    const uint64_t R1 = m_ne64(R & reach, 0ull);
                    R = m_ne64(R1 & reach, 0ull);
                    if (R == R1) break;
                }
                const uint64_t New = R & ~cur.V;
                if (New != 0ull) {
                    uint64_t was = 0ull;
                    if constexpr (kMulti) {
                        uint32_t slot_c;
                        uint64_t claimed;
                        (void)S.lookup(tile | kSrcClaim, slot_c, claimed);  // (find or insert)
                        if (lane == 0) {
                            was = S.add_walked(slot_c, New);     // claimed before by whichever source
                            (void)S.add_walked(cur.slot, New);   // seen by this one, claimed or not
                        }
                    } else {
                        if (lane == 0) was = S.add_walked(cur.slot, New);
                    }
                    const uint64_t first_here = New & ~uni64(was);  // the pixels this step was the first to walk
                    if (first_here != 0ull) {
                        const uint64_t H = Rg & m_ne64(first_here & adj, 0ull);
                        if (H != 0ull) team_push8(S, tile, H, lane, pc, tag);
                    }
                }
            }
            if (lane == 0) S.mark_processed(i);
        }
        __syncthreads();
        if (threadIdx.x == 0) sh->end = sh->tail;
        __syncthreads();
        const uint32_t ne = sh->end;
        if (sh->overflow != 0u || ++levels > kMaxSteps) {
            rc = 1;
            break;
        }
        if (ne == ge) break;
        gb = ge;
        ge = ne;
    }
    *begin_out = gb;
    *steps_out = steps;
    return rc;
}

// (The same walk WITHOUT levels -- the ring as a queue: a wavefront claims the record at the head by compare-and-swap, waits
// for its slot to be written, walks the tile, appends, comes back; head and the number of record holders in one word, so
// that "queue empty and nobody holds a record" is one read -- was built, is exact, and is slower: natural 4K frame 1.67 ms
// of flood against 1.51, and walks handed over at 160 tiles still cost the synthetic frames 0.1-0.6 ms
// (profiles/r03_flood_team_queue_variant.txt): a claimed step is 3-4 us -- poll, claim, slot, lookup, two atomic ORs, append,
// all LDS round trips in a chain -- against 1.4 us for a step of the single-wavefront walk, which keeps that chain in
// registers.  A first version counted the wavefronts "trying to claim" in a counter of its own: three and more idle
// wavefronts then kept each other waiting for ever, each finding another in the middle of its attempt.  Not in the tree.)
constexpr size_t kTeamLdsBytes = (size_t)(3 * kRingTeam + 3 * kHashBig + 2 * kPend) * 4 + (size_t)kHashBig * 2 + sizeof(TeamShared);
// from_multi_list: the list is the round's way-point seeds (A.multi_list, written by the last survivors pass) instead of
// what this round's first tier handed over; nothing comes with its entries.
__global__ __launch_bounds__(64 * kTeamWaves) void flood_explore_team_kernel(FloodArgs A, BinTrig trig,
                                                                             const uint32_t* __restrict__ big_list,
                                                                             uint32_t from_multi_list) {
    extern __shared__ uint32_t s_team[];
    const int lane = threadIdx.x & 63, wave = (int)uni(threadIdx.x >> 6);
    if (uni(A.ctrl[kCtrlNAct]) == 0u || giant_pending(A)) return;  // a round enqueued past the end (or past a stall: the listed seeds are the ordered tail's)
    const uint32_t n_big_raw = uni(A.ctrl[from_multi_list ? kCtrlNMulti : kCtrlNBig]);
    const uint32_t n_big = from_multi_list ? min(n_big_raw, kBigCap) : (n_big_raw < A.big_cap ? n_big_raw : A.big_cap);
    uint32_t* ring = s_team;
    uint32_t* hash = ring + 3 * kRingTeam;
    uint32_t* pend = hash + 3 * kHashBig;
    uint16_t* ord = reinterpret_cast<uint16_t*>(pend + 2 * kPend);
    TeamShared* sh = reinterpret_cast<TeamShared*>(reinterpret_cast<char*>(ord) + (size_t)kHashBig * 2);
    TeamStore S{ring, ring + kRingTeam, ring + 2 * kRingTeam, hash, hash + kHashBig, hash + 2 * kHashBig, ord, sh};
    for (uint32_t ai = uni(blockIdx.x); ai < n_big; ai += gridDim.x) {
        const uint32_t k = uni(big_list[ai]);
        // (a listed seed above the round's window is not walked at all, like every seed there: a commit is only valid if
        // every lower active seed has walked)
        if (from_multi_list && k >= uni(A.ctrl[kCtrlWindow])) continue;
        const int s = (int)uni((uint32_t)A.seed_idx[k]);
        const int b = (int)uni((uint32_t)A.seed_bin[k]);
        const float thr = __uint_as_float(uni(__float_as_uint(A.seed_thr[k])));
        const float sn = trig.st[b], cs = trig.ct[b];
        // (the checks of explore_seed again: the first tier passed them in this round, on the same labels)
        const uint32_t seed_label = uni(A.label[s]);
        const bool own = seed_label == k;
        if (seed_label < kMarkBit && !own) continue;
        if (!own && !(((uni((uint32_t)A.dmask[s]) >> b) & 1u) &&
                      directional(__uint_as_float(uni(__float_as_uint(A.dx[s]))), __uint_as_float(uni(__float_as_uint(A.dy[s]))), sn, cs) > thr)) {
            if (threadIdx.x == 0) A.flags[k] = kFlagSelfFail;
            continue;
        }
        // A frame that has held back giant_many walks already is one of overlapping giants (a ramp under noise: thousands of
        // weak seeds that each reach 100 000 pixels until the stronger ones have committed), and walking the table full to
        // find the next one out is most of its first round (4K ramp: 3.2 of 4.3 ms).  From then on the walks of the weaker
        // three quarters of the seeds are not even begun: they count as unfinished (nothing above them commits), and the end
        // of the round closes the window in front of them -- kCtrlStaged: the strongest quarter first, twice as many a round.
        // (only in a round whose end will close the window: flood_advance tests the same words)
        if (A.giant_hold != 0u && k != uni(A.ctrl[kCtrlLowest]) &&
            ((k >= (uni(A.ctrl[kCtrlNSeeds]) >> 2) && giants_many(A.ctrl, A.giant_many)) || (!from_multi_list && giants_all(A.ctrl)))) {
            if (threadIdx.x == 0) {
                A.flags[k] = kFlagIncomplete;
                atomicMin(&A.ctrl[kCtrlBarrier], k);
            }
            continue;
        }
        // what the first tier handed over with this entry (A.big_cap entries at most, so ai is its position in the list)
        const uint32_t* hb = A.handover + (size_t)ai * kFloodHandWords;
        const uint32_t h_recs_in = from_multi_list ? 0u : min(uni(hb[1]), kHandRecs);
        // A seed that left way-points on its footprint (save_waypoints) and comes without a walk in progress is walked from
        // all of them at once (team_walk, kMulti).  Should the table run out -- two entries a tile -- it starts again, plainly.
        const uint32_t wp_hdr = (A.wp_min_tiles != 0xFFFFFFFFu && k < A.wp_cap) ? uni(A.waypoints[(size_t)k * kFloodWpWords]) : 0u;
        const bool multi = h_recs_in == 0u && wp_hdr != 0u && (wp_hdr >> 8) <= kWpMaxTiles;
        uint32_t begin = 0u, my_steps = 0u;
        int rc = 0;
        bool did_multi = false;
        if (multi) {
            __syncthreads();  // the previous seed's table is no longer read
            for (int i = (int)threadIdx.x; i < kHashBig; i += 64 * kTeamWaves) {
                S.hk[i] = 0u;
                S.hv0[i] = 0u;
                S.hv1[i] = 0u;
            }
            if (threadIdx.x <= kWpK) {  // source 0: the seed's own pixel; 1 .. kWpK: the way-points
                const uint32_t t = threadIdx.x;
                uint32_t q = t == 0u ? (uint32_t)s : A.waypoints[(size_t)k * kFloodWpWords + t];
                uint64_t m = 0ull;
                if (q == kWpNone || q >= (uint32_t)A.w * (uint32_t)A.h) q = (uint32_t)s;  // (no way-point here: an empty record)
                else m = 1ull << (((q / (uint32_t)A.w) & 7u) * 8u + ((q % (uint32_t)A.w) & 7u));
                const uint32_t qr = q / (uint32_t)A.w, qc = q % (uint32_t)A.w;
                S.rt[t] = (((qr >> 3) << 16) | (qc >> 3)) | (t << kSrcShift);
                S.rlo[t] = (uint32_t)m;
                S.rhi[t] = (uint32_t)(m >> 32);
                sh->adj[t] = 0u;
            }
            if (threadIdx.x == 0) {
                sh->tail = kWpK + 1u;
                sh->end = kWpK + 1u;
                sh->ntiles = 0u;
                sh->blocked = 0u;
                sh->blocker = 0xFFFFFFFFu;
                sh->overflow = 0u;
                sh->cnt = 0u;
                sh->steps = 0u;
                sh->reach = 0u;
                sh->ctiles = 0u;
            }
            __syncthreads();
            uint32_t multi_steps = 0u;
            rc = team_walk<TeamStore, true>(A, k, b, thr, sn, cs, S, lane, wave, own, kWpK + 1u, A.team_tiles, &begin, &multi_steps);
            my_steps += multi_steps;
            did_multi = rc == 0;
        }
        if (!did_multi) {
            __syncthreads();  // the previous seed's table is no longer read
            for (int i = (int)threadIdx.x; i < kHashBig; i += 64 * kTeamWaves) {
                S.hk[i] = 0u;
                S.hv0[i] = 0u;
                S.hv1[i] = 0u;
            }
            // what the first tier handed over with this entry (A.big_cap entries at most, so ai is its position in the list)
            const uint32_t* hb = A.handover + (size_t)ai * kFloodHandWords;
            const uint32_t h_tiles = min(uni(hb[0]), kHandTiles), h_recs = h_recs_in;
            const uint32_t first_level = h_recs ? h_recs : 1u;
            __syncthreads();  // (table cleared)
            if (h_recs) {
                for (uint32_t t = threadIdx.x; t < h_tiles; t += 64u * kTeamWaves) {
                    const uint32_t key = hb[kHandTable + 3u * t];
                    uint32_t hs = (key * 2654435761u) >> TeamStore::kHashShift;
                    while (atomicCAS(&S.hk[hs], 0u, key) != 0u) hs = (hs + 1u) & (uint32_t)(kHashBig - 1);  // (the tiles are distinct)
                    S.hv0[hs] = hb[kHandTable + 3u * t + 1u];
                    S.hv1[hs] = hb[kHandTable + 3u * t + 2u];
                    S.ord[t] = (uint16_t)hs;
                }
                for (uint32_t t = threadIdx.x; t < h_recs; t += 64u * kTeamWaves) {
                    S.rt[t] = hb[kHandRing + 3u * t];
                    S.rlo[t] = hb[kHandRing + 3u * t + 1u];
                    S.rhi[t] = hb[kHandRing + 3u * t + 2u];
                }
            }
            if (threadIdx.x == 0) {
                if (!h_recs) {
                    const int sr = s / A.w, sc = s - sr * A.w;
                    const uint64_t m = 1ull << ((sr & 7) * 8 + (sc & 7));
                    S.rt[0] = ((uint32_t)(sr >> 3) << 16) | (uint32_t)(sc >> 3);
                    S.rlo[0] = (uint32_t)m;
                    S.rhi[0] = (uint32_t)(m >> 32);
                }
                sh->tail = first_level;
                sh->end = first_level;
                sh->ntiles = h_recs ? h_tiles : 0u;
                sh->blocked = 0u;
                sh->blocker = 0xFFFFFFFFu;
                sh->overflow = 0u;
                sh->cnt = 0u;
                sh->steps = h_recs ? hb[2] : 0u;
            }
            __syncthreads();
            rc = team_walk(A, k, b, thr, sn, cs, S, lane, wave, own, first_level, A.team_tiles, &begin, &my_steps);
        }
        // (team_walk ends behind a barrier: every wavefront sees the final table)
        WalkState st{0u, 0u, 0u, 0u, false, 0u, 0u};
        uint32_t px = 0u;
        bool in_slab = false;
        if (rc != 0 && A.giant_hold != 0u && (A.giant_step != 0u || k != uni(A.ctrl[kCtrlLowest]))) {
            // LDS exhausted, and this is not the lowest active seed: a giant, held back (see kCtrlLowest).  Nothing has been
            // stamped; the seed counts as a walk that did not finish, so nothing above it commits in this round.  (With the
            // giant step the lowest active seed is marked as well, and the end of the round asks for the step: flood_advance.)
            if (threadIdx.x == 0) {
                A.tier[k] = (uint8_t)(A.tier[k] | 4u);
                A.flags[k] = kFlagIncomplete;
                atomicMin(&A.ctrl[kCtrlBarrier], k);
                atomicAdd(&A.ctrl[kCtrlGiants], 1u);
                if (!from_multi_list) {
                    atomicAdd(&A.ctrl[kCtrlTeamGiants], 1u);
                    atomicAdd(&A.ctrl[kCtrlTeamDone], 1u);
                }
            }
            continue;
        }
        if (rc != 0) {
            // LDS exhausted.  The whole team moves into a global slab and goes on there (TeamGlobalStore), from the records
            // that are still unprocessed; nothing has been stamped yet, the table travels.
            if (threadIdx.x == 0) {
                const uint32_t slab = atomicAdd(&A.ctrl[kCtrlSlabs], 1u);
                if (slab < A.n_slabs) atomicAdd(&A.ctrl[kCtrlSlabTotal], 1u);
                sh->pad = slab;
            }
            __syncthreads();
            const uint32_t slab = sh->pad;
            if (slab < A.n_slabs) {
                in_slab = true;
                TeamGlobalStore G{A.slab_ring + (size_t)slab * A.slab_ring_cap, A.slab_hash + (size_t)slab * A.slab_hash_cap * 2,
                                  A.slab_ring_cap, A.slab_hash_cap, sh};
                for (uint32_t i = threadIdx.x; i < 2u * A.slab_hash_cap; i += 64u * kTeamWaves) G.hash[i] = make_uint4(0u, 0u, 0u, 0u);
                const uint32_t n_tiles = sh->ntiles, tail = sh->tail;
                __syncthreads();  // (the cleared table is in L2; sh->cnt below is used as the compaction counter)
                for (uint32_t t = threadIdx.x; t < n_tiles; t += 64u * kTeamWaves) {
                    const uint32_t slot = S.ord[t], key = S.hk[slot];
                    uint32_t hs = G.hash_of(key);
                    while (atomicCAS(TeamGlobalStore::w(&G.hash[2u * hs], 1), 0u, key) != 0u) hs = (hs + 1u) & G.hash_mask();
                    TeamGlobalStore::stw(&G.hash[2u * hs], 2, S.hv0[slot]);
                    TeamGlobalStore::stw(&G.hash[2u * hs], 3, S.hv1[slot]);
                    TeamGlobalStore::stw(&G.hash[2u * t + 1u], 0, hs);
                }
                for (uint32_t i = begin + threadIdx.x; (int32_t)(tail - i) > 0; i += 64u * kTeamWaves) {
                    const uint32_t j = i & (uint32_t)(kRingTeam - 1);
                    const uint32_t t = S.rt[j];
                    if (t == kVoidTile) continue;
                    G.write_record(atomicAdd(&sh->cnt, 1u), t, ((uint64_t)S.rhi[j] << 32) | S.rlo[j]);
                }
                __syncthreads();
                const uint32_t first = sh->cnt;
                __syncthreads();
                if (threadIdx.x == 0) {
                    sh->tail = first;
                    sh->end = first;
                    sh->overflow = 0u;
                    sh->cnt = 0u;
                }
                __syncthreads();
                uint32_t more_steps = 0u;
                rc = first ? team_walk(A, k, b, thr, sn, cs, G, lane, wave, own, first, 0xFFFFFFFFu, &begin, &more_steps) : 0;
                my_steps += more_steps;
                // stamps and pixel count from the slab's table, the tiles shared out as in stamp_footprint
                const uint32_t mine = kMarkBit | k, nt = sh->ntiles;
                const int lr = lane >> 3, lc = lane & 7;
                bool foreign = false;
                for (uint32_t i = (uint32_t)wave; i < nt; i += kTeamWaves) {
                    const uint32_t slot = uni(G.ord_slot(i));
                    const uint32_t tile = uni(G.peek_key(slot)) - 1u;
                    const uint64_t V = uni64(G.peek_walked(slot));
                    if (lane == 0) px += (uint32_t)__popcll(V);
                    if ((V >> lane) & 1ull) {
                        const size_t q = (size_t)((tile >> 16) * 8 + lr) * A.w + ((tile & 0xFFFFu) * 8 + lc);
                        const uint32_t old = atomicMin(&A.label[q], mine);
                        A.dirty[q >> 8] = 1;
                        if (old > mine) {
                            if (old != kLabelFree) A.blocked[old & ~kMarkBit] = 1u;
                        } else if (old < mine && old >= kMarkBit) {
                            foreign = true;
                        }
                    }
                }
                if (__ballot(foreign)) st.blocked = true;  // (no blocker noted: the lowest active seed's walk, or a test hook's)
                st.ntiles = nt;
            }
        }
        if (did_multi) {
            // Which sources belong to the seed?  Two sources that have seen a common pixel are joined (see team_walk); the
            // footprint is what the sources joined with source 0 have seen.
            for (int i = (int)threadIdx.x; i < kHashBig; i += 64 * kTeamWaves) {
                const uint32_t key = S.hk[i];
                if (key == 0u) continue;
                const uint32_t tk = key - 1u, src = tk >> kSrcShift;
                if (src == 0u || src > kWpK) continue;  // (source 0 is met from the other side; claim entries join nobody)
                const uint64_t V = S.peek_walked((uint32_t)i);
                if (V == 0ull) continue;
                for (uint32_t s2 = 0u; s2 < src; ++s2) {
                    uint32_t sl;
                    if (S.find_ro((tk & ~kSrcMask) | (s2 << kSrcShift), sl) && (S.peek_walked(sl) & V) != 0ull) {
                        atomicOr(&sh->adj[src], 1u << s2);
                        atomicOr(&sh->adj[s2], 1u << src);
                    }
                }
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                uint32_t reach = 1u;
                for (;;) {
                    uint32_t nr = reach;
                    for (uint32_t s2 = 0u; s2 <= kWpK; ++s2)
                        if ((reach >> s2) & 1u) nr |= sh->adj[s2];
                    if (nr == reach) break;
                    reach = nr;
                }
                sh->reach = reach;
                atomicAdd(&A.ctrl[kCtrlMulti], 1u);
            }
            __syncthreads();
            // Stamps and pixel count, tile by tile over the claim entries (one per tile of whatever any source walked): the
            // tile's share of the footprint is what the joined sources have seen of it, looked up by lanes 0 .. kWpK side by side.
            const uint32_t reach = sh->reach, nt = sh->ntiles, mine = kMarkBit | k;
            const int lr = lane >> 3, lc = lane & 7;
            bool foreign = false;
            uint32_t ctl = 0u;
            for (uint32_t i = (uint32_t)wave; i < nt; i += kTeamWaves) {
                const uint32_t slot = uni((uint32_t)S.ord[i]);
                const uint32_t tk = uni(S.hk[slot]) - 1u;
                if ((tk & kSrcMask) != kSrcClaim) continue;
                const uint32_t T = tk & ~kSrcMask;
                uint32_t u0 = 0u, u1 = 0u;
                if ((uint32_t)lane <= kWpK && ((reach >> lane) & 1u)) {
                    uint32_t sl;
                    if (S.find_ro(T | ((uint32_t)lane << kSrcShift), sl)) {
                        u0 = S.hv0[sl];
                        u1 = S.hv1[sl];
                    }
                }
#pragma unroll
                for (int off = 4; off >= 1; off >>= 1) {
                    u0 |= (uint32_t)__shfl_xor((int)u0, off);
                    u1 |= (uint32_t)__shfl_xor((int)u1, off);
                }
                const uint64_t U = uni64(u0, u1);
                if (U == 0ull) continue;
                ctl += 1u;
                if (lane == 0) px += (uint32_t)__popcll(U);
                if ((U >> lane) & 1ull) {
                    const size_t q = (size_t)((T >> 16) * 8 + lr) * A.w + ((T & 0xFFFFu) * 8 + lc);
                    const uint32_t old = atomicMin(&A.label[q], mine);
                    A.dirty[q >> 8] = 1;
                    if (old > mine) {
                        if (old != kLabelFree) A.blocked[old & ~kMarkBit] = 1u;
                    } else if (old < mine && old >= kMarkBit) {
                        foreign = true;
                    }
                }
            }
            if (__ballot(foreign)) st.blocked = true;
            if (lane == 0 && ctl) atomicAdd(&sh->ctiles, ctl);
        } else if (!in_slab) {
            st.ntiles = sh->ntiles;
            stamp_footprint(A, k, S, st, lane, (uint32_t)wave * 8u, 8u * kTeamWaves);
            for (int i = (int)threadIdx.x; i < kHashBig; i += 64 * kTeamWaves) px += (uint32_t)__popc(S.hv0[i]) + (uint32_t)__popc(S.hv1[i]);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) px += (uint32_t)__shfl_xor((int)px, off);
        if (lane == 0) {
            atomicAdd(&sh->cnt, px);
            atomicAdd(&sh->steps, my_steps);
            if (st.blocked) atomicOr(&sh->blocked, 1u);
            if (st.blocker != 0xFFFFFFFFu) atomicMin(&sh->blocker, st.blocker);
        }
        __syncthreads();
        if (wave != 0) continue;  // (the first wavefront finishes the seed; the others wait at the next seed's barrier or leave)
        st.cnt = sh->cnt;
        st.steps = sh->steps;
        st.blocked = sh->blocked != 0u;
        st.blocker = sh->blocker;
        if (did_multi) st.ntiles = sh->ctiles;
        // a long walk that came here for the first time (handed over by the first tier) leaves its way-points now
        if (!did_multi && !in_slab && rc == 0 && wp_hdr == 0u && k < A.wp_cap && st.ntiles >= A.wp_min_tiles && st.ntiles <= kWpMaxTiles &&
            st.cnt <= kWpThinPx * st.ntiles)
            save_waypoints(A, k, S, st.ntiles, lane);
        // a marked giant whose footprint fits by now is an ordinary seed again
        if (rc == 0 && lane == 0 && (A.tier[k] & 4u)) A.tier[k] = (uint8_t)(A.tier[k] & ~4u);
        // ... and its records (save_log), for flood_rewalk_kernel
        if (!did_multi && !in_slab && rc == 0 && st.ntiles >= A.log_min_tiles && st.ntiles <= A.log_max_len && k < A.log_seeds &&
            uni(A.log_len[k]) == 0u)
            save_log(A, k, S, st.ntiles, lane);
        if (rc != 0 && lane == 0) {  // no slab to go to, or the slab ran out as well: the ordered tail will finish this seed
            A.flags[k] = kFlagIncomplete;
            atomicMin(&A.ctrl[kCtrlBarrier], k);
        }
        // A long walk of a WEAK seed that ends blocked by lower seeds it knows, and that has no log to turn to next time, is
        // walked again every round for as long as that seed is unresolved -- one team, tile level after tile level, while the
        // chain of small seeds below it resolves one a round (a frame of soft blobs: thirteen rounds of 0.45 ms for seed 46 046's
        // 2 300 steps).  It waits instead: the survivors pass closes the window in front of the lowest such seed until its
        // blocker is resolved (kCtrlDeferLow; any prefix of the seed order is a valid window).  Only for the weaker half of
        // the seeds: a window closed in front of a strong seed would keep the whole frame waiting.
        if (lane == 0 && A.blk != nullptr) {
            const bool no_log = !(k < A.log_seeds && A.log_min_tiles != 0xFFFFFFFFu && A.log_len[k] != 0u);
            // (the LOWEST of the lower seeds it met.  Waiting for the highest -- the chain below it resolves from its lowest seed
            // up, and the walk finds the next of them in its way every other round -- was measured and is no better: the window
            // stays closed longer and the seeds above pay with rounds of their own, 33 instead of 22 on the frame of soft blobs)
            A.blk[k] = (rc == 0 && st.blocked && st.blocker < k && st.steps >= A.defer_steps && k >= (A.ctrl[kCtrlNSeeds] >> 1) && no_log) ? st.blocker : 0xFFFFFFFFu;
        }
        if (lane == 0) {
            A.count[k] = st.cnt;
            if (st.blocked) A.blocked[k] = 1u;
            A.flags[k] |= st.steps << 8;
            // (what makes a frame "regional", explore_body: walks the first tier could not have held, however early they
            // were handed over)
            if (st.ntiles > kHandTiles) atomicAdd(&A.ctrl[kCtrlBigLong], 1u);
            if (!from_multi_list) atomicAdd(&A.ctrl[kCtrlTeamDone], 1u);  // (giants_all)
        }
    }
}

// ---- Re-walks from the log ---------------------------------------------------------------------------------------------------
// A footprint is the connected set around the seed in {acceptable, not committed}, acceptance is static, and what borders a
// footprint is unacceptable or committed for good: a seed's footprint can only shrink, and its next footprint is the
// connected part around the seed of (ANY earlier footprint of it, minus the pixels committed since).  A walk is a chain of
// dependent steps, one tile each, ~1.4 us a step: rounds 2-5 of the 4K bench frame lasted 134, 108, 233, 232 us for their
// one longest re-walk (142-183 tiles) while the chip idled.  So a finished walk of log_min_tiles tiles or more leaves its
// (tile, walked pixels) records (save_log), and in the seed's later rounds
//  - the exploration walks log_walk_tiles tiles of it (most footprints have shrunk to a handful of tiles by their second
//    round, and a short walk is cheaper than the records of a long one); if that is not the end of the walk the seed goes
//    on the round's list (multi_list; a log that the last round has cut down already and is still long goes there at once),
//  - this kernel, launched behind the exploration, works the footprint out from the records, a workgroup a seed: the
//    records' pixels are looked at once (all tiles' loads in flight together -- no chain), the components of each tile's
//    surviving pixels become nodes (a thread a tile, 64-bit masks), pairs of nodes of neighbouring tiles that touch are
//    noted and then united (lock-free union-find in LDS: CAS on roots, path halving), and the pixels of the nodes united
//    with the seed's node are the footprint.  Stamps, counts and blocked marks are what a walk would have left; the log is
//    rewritten with the new footprint's records only.
// Logs the tables cannot take (more than kRwComps components in a tile: noise; more nodes or pairs than there is room for)
// go a slower way in the same launch: sweeps over the tiles' reached sets until nothing changes (see there).
// Second-tier walks (up to 2048 tiles) leave logs too when the frame is expected to have them; a second instance of the
// kernel with larger tables works on those.  Single 4K frames: flood 1.31 -> 0.92 ms (profiles/r04_flood_logs.txt).
constexpr int kRwComps = 6;
constexpr uint32_t kRwBatch = 16;  // tiles a wavefront has in flight (the pixels of a log are looked at, and stamped, with no chain between them)
constexpr uint64_t kCol0 = 0x0101010101010101ull, kCol7 = 0x8080808080808080ull;
__device__ __forceinline__ uint64_t dilate8(uint64_t m) {
    const uint64_t hz = m | ((m << 1) & ~kCol0) | ((m >> 1) & ~kCol7);
    return hz | (hz << 8) | (hz >> 8);
}
__device__ __forceinline__ uint32_t rw_find(volatile uint32_t* par, uint32_t x) {
    for (;;) {
        const uint32_t p = par[x];
        if (p == x) return x;
        const uint32_t g = par[p];
        if (g != p) par[x] = g;  // (any ancestor is a valid parent: a lost or stale write only costs a step)
        x = g;
    }
}
// (Roots are linked in a scrambled order of the node numbers, not the numbers' own: the nodes of a line are numbered along
// the line, and "larger under smaller" strings them into one chain as deep as the line is long -- every find then walks it.)
__device__ __forceinline__ uint32_t rw_order(uint32_t x) { return x * 0x9E3779B1u; }
__device__ __forceinline__ void rw_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        uint32_t ra = rw_find(par, a), rb = rw_find(par, b);
        if (ra == rb) return;
        if (rw_order(ra) < rw_order(rb)) {
            const uint32_t t = ra;
            ra = rb;
            rb = t;
        }
        if (atomicCAS(&par[ra], ra, rb) == ra) return;  // (a root only ever changes through this exchange)
    }
}
constexpr int kRewalkTiles = 256;       // records a single wavefront's tables hold (the first tier's walks: 192 tiles at most)
constexpr uint32_t kRewalkGrid = 2048;  // workgroups of a launch (they stride over the list: round two of a 4K frame has ~1 800 entries; empty workgroups cost the dispatcher)
constexpr int kRewalkThreads = 256;
constexpr int kRewalkTilesBig = 2048, kRewalkThreadsBig = 1024;  // the second tier's walks (its team's table: 1536 tiles)
template <int kTiles>
constexpr size_t rewalk_lds_bytes() {
    // records 3 words, table keys 2, nodes 2 x 3, pairs 4 words a tile; table indices 2, first node 1 half-words; components 1 byte
    return (size_t)kTiles * 3 * 4 + (size_t)2 * kTiles * 4 + (size_t)2 * kTiles * 3 * 4 + (size_t)4 * kTiles * 4 + (size_t)2 * kTiles * 2 +
           (size_t)kTiles * 2 + (size_t)kTiles;
}
#ifdef LR_REWALK_TIMING
// Diagnostic build (LR_EXTRA_FLAGS=-DLR_REWALK_TIMING, printed by LIBRECTIFY_FLOOD_DEBUG): [0] seeds, [1] records, then
// s_memtime ticks of thread 0: [2] records in, [3] pixels looked at, [4] components, [5] unions, [6] footprint, [7] stamps.
__device__ unsigned long long g_rewalk_timing[8];
#define RW_TICK(i)                                            \
    if (threadIdx.x == 0) {                                   \
        const uint64_t t_ = __builtin_amdgcn_s_memtime();     \
        atomicAdd(&g_rewalk_timing[i], (unsigned long long)(t_ - tlast)); \
        tlast = t_;                                           \
    }
#else
#define RW_TICK(i)
#endif
template <int kThreads, int kTiles>
__global__ __launch_bounds__(kThreads) void flood_rewalk_kernel(FloodArgs A, const uint32_t* __restrict__ list, uint32_t min_len) {
    constexpr int kNodes = 2 * kTiles, kHash = 2 * kTiles, kWaves = kThreads / 64, kEdges = 4 * kTiles;
    static_assert(kNodes <= 65536, "node pairs are packed into 32 bits");
    constexpr int kHashShift = 32 - __builtin_ctz((unsigned)kHash);
    extern __shared__ uint32_t s_rw[];
    uint32_t* tid = s_rw;            // per record: tile,
    uint32_t* mlo = tid + kTiles;    // pixels (walked; then: walked and still there; then: the new footprint's)
    uint32_t* mhi = mlo + kTiles;
    uint32_t* hkey = mhi + kTiles;   // tile + 1 -> record (open addressing)
    uint32_t* nlo = hkey + kHash;    // per node: pixels,
    uint32_t* nhi = nlo + kNodes;
    uint32_t* par = nhi + kNodes;    // parent
    uint32_t* edge = par + kNodes;   // pairs of touching nodes
    uint16_t* hidx = reinterpret_cast<uint16_t*>(edge + kEdges);
    uint16_t* tfirst = hidx + kHash;
    uint8_t* tcomp = reinterpret_cast<uint8_t*>(tfirst + kTiles);
    __shared__ uint32_t s_nnodes, s_over, s_blocked, s_cnt, s_seed_node, s_nout, s_nedges;
    const int lane = threadIdx.x & 63, wave = (int)uni(threadIdx.x >> 6);
    const int lr = lane >> 3, lc = lane & 7;
    if (uni(A.ctrl[kCtrlNAct]) == 0u || giant_pending(A)) return;  // a round enqueued past the end, or past a stall
    const uint32_t n_list = min(uni(A.ctrl[kCtrlNMulti]), kBigCap);
    const uint32_t window = uni(A.ctrl[kCtrlWindow]);
    for (uint32_t ai = uni(blockIdx.x); ai < n_list; ai += gridDim.x) {
        const uint32_t k = uni(list[ai]);
        if (k >= window) continue;  // (not walked at all, like every seed above the window)
        const uint32_t n = uni(A.log_len[k]) & ~kLogShrunk;
        if (n < min_len || n > (uint32_t)kTiles) continue;  // (another instance of this kernel takes it)
        const uint32_t s = uni((uint32_t)A.seed_idx[k]);
        const uint32_t seed_label = uni(A.label[s]);
        const bool own = seed_label == k;
        if (seed_label < kMarkBit && !own) continue;  // dead: the survivors pass takes it off the list
        const uint32_t mine = kMarkBit | k;
        uint32_t* rec = A.log_buf + (size_t)uni(A.log_off[k]) * 3u;
        __syncthreads();  // (the previous seed's tables are no longer read)
#ifdef LR_REWALK_TIMING
        uint64_t tlast = __builtin_amdgcn_s_memtime();
        if (threadIdx.x == 0) {
            atomicAdd(&g_rewalk_timing[0], 1ull);
            atomicAdd(&g_rewalk_timing[1], (unsigned long long)n);
        }
#endif
        for (uint32_t i = threadIdx.x; i < n; i += kThreads) {
            tid[i] = rec[3u * i];
            mlo[i] = rec[3u * i + 1u];
            mhi[i] = rec[3u * i + 2u];
        }
        for (uint32_t i = threadIdx.x; i < (uint32_t)kHash; i += kThreads) hkey[i] = 0u;
        if (threadIdx.x == 0) {
            s_nnodes = 0u;
            s_over = 0u;
            s_blocked = 0u;
            s_cnt = 0u;
            s_seed_node = 0xFFFFFFFFu;
            s_nout = 0u;
            s_nedges = 0u;
        }
        __syncthreads();
        RW_TICK(2)
        // (1) which of the logged pixels are still there: free, stamped by this round's walks, or this seed's own ground
        for (uint32_t i0 = (uint32_t)wave * kRwBatch; i0 < n; i0 += (uint32_t)kWaves * kRwBatch) {
            uint32_t lab[kRwBatch];
            bool in[kRwBatch];
#pragma unroll
            for (int j = 0; j < (int)kRwBatch; ++j) {
                const uint32_t i = i0 + (uint32_t)j;
                lab[j] = 0u;
                in[j] = false;
                if (i < n) {
                    const uint32_t tile = tid[i];
                    const uint64_t V = ((uint64_t)mhi[i] << 32) | mlo[i];
                    in[j] = (V >> lane) & 1ull;
                    if (in[j]) lab[j] = A.label[(size_t)((tile >> 16) * 8 + lr) * A.w + ((tile & 0xFFFFu) * 8 + lc)];
                }
            }
#pragma unroll
            for (int j = 0; j < (int)kRwBatch; ++j) {
                const uint32_t i = i0 + (uint32_t)j;
                const uint64_t M = __ballot(in[j] && (lab[j] >= kMarkBit || lab[j] == k));
                if (i < n && lane == 0) {
                    mlo[i] = (uint32_t)M;
                    mhi[i] = (uint32_t)(M >> 32);
                }
            }
        }
        __syncthreads();
        RW_TICK(3)
        // (2) a thread a record: the tile goes into the table, the components of its pixels become nodes
        const uint32_t seed_tile = ((s / (uint32_t)A.w) >> 3) << 16 | ((s % (uint32_t)A.w) >> 3);
        const uint64_t seed_bit = 1ull << (((s / (uint32_t)A.w) & 7u) * 8u + ((s % (uint32_t)A.w) & 7u));
        for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
            const uint32_t key = tid[t] + 1u;
            uint32_t hs = (key * 2654435761u) >> kHashShift;
            while (atomicCAS(&hkey[hs], 0u, key) != 0u) hs = (hs + 1u) & (uint32_t)(kHash - 1);  // (the records' tiles are distinct)
            hidx[hs] = (uint16_t)t;
            uint64_t rem = ((uint64_t)mhi[t] << 32) | mlo[t];
            uint64_t comp[kRwComps];
            int nc = 0;
            bool over = false;
            while (rem != 0ull) {
                uint64_t c = rem & (~rem + 1ull), prev;
                do {
                    prev = c;
                    c = dilate8(c) & rem;
                } while (c != prev);
                rem &= ~c;
                if (nc == kRwComps) {
                    over = true;
                    break;
                }
                comp[nc++] = c;
            }
            uint32_t first = 0u;
            if (nc) first = atomicAdd(&s_nnodes, (uint32_t)nc);
            if (over || first + (uint32_t)nc > (uint32_t)kNodes) {
                s_over = 1u;
                nc = 0;
            }
            tfirst[t] = (uint16_t)first;
            tcomp[t] = (uint8_t)nc;
            for (int c = 0; c < nc; ++c) {
                nlo[first + c] = (uint32_t)comp[c];
                nhi[first + c] = (uint32_t)(comp[c] >> 32);
                par[first + c] = first + (uint32_t)c;
                if (tid[t] == seed_tile && (comp[c] & seed_bit)) s_seed_node = first + (uint32_t)c;
            }
        }
        __syncthreads();
        RW_TICK(4)
        // (tables too small for this log -- more than kRwComps components in a tile, more nodes than the table holds: the
        // footprint is worked out by sweeps instead, below)
        bool sweep = uni(s_over) != 0u || uni(s_seed_node) == 0xFFFFFFFFu || A.log_sweep != 0u;
        if (!sweep) {
        // (3) nodes of neighbouring tiles that touch are one: east, south, south-east, south-west of every tile
        for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
            const int nc = tcomp[t];
            if (nc == 0) continue;
            const uint32_t tile = tid[t], first = tfirst[t];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const uint32_t key = tile + (d == 0 ? 1u : d == 1 ? 0x10000u : d == 2 ? 0x10001u : 0xFFFFu) + 1u;
                uint32_t hs = (key * 2654435761u) >> kHashShift, cur;
                while ((cur = hkey[hs]) != key && cur != 0u) hs = (hs + 1u) & (uint32_t)(kHash - 1);
                if (cur != key) continue;
                const uint32_t u = hidx[hs];
                const int ncu = tcomp[u];
                const uint32_t ufirst = tfirst[u];
                for (int c = 0; c < nc; ++c) {
                    const uint64_t a = ((uint64_t)nhi[first + c] << 32) | nlo[first + c];
                    uint64_t e;  // the pixels of the neighbouring tile that touch this node
                    if (d == 0) {
                        e = (a >> 7) & kCol0;
                        e |= (e << 8) | (e >> 8);
                    } else if (d == 1) {
                        e = a >> 56;
                        e = (e | (e << 1) | (e >> 1)) & 0xFFull;
                    } else if (d == 2) {
                        e = a >> 63;
                    } else {
                        e = ((a >> 56) & 1ull) << 7;
                    }
                    if (e == 0ull) continue;
                    for (int cu = 0; cu < ncu; ++cu) {
                        const uint64_t bmask = ((uint64_t)nhi[ufirst + cu] << 32) | nlo[ufirst + cu];
                        if (e & bmask) {
                            // (the pair is only noted here: a union is a chain of dependent LDS round trips, and inside these
                            // divergent loops the wavefront would pay it once per (direction, node, node) combination)
                            const uint32_t pos = atomicAdd(&s_nedges, 1u);
                            if (pos < (uint32_t)kEdges) edge[pos] = ((first + (uint32_t)c) << 16) | (ufirst + (uint32_t)cu);
                        }
                    }
                }
            }
        }
        __syncthreads();
        const uint32_t n_edges = uni(s_nedges);
        if (n_edges > (uint32_t)kEdges) sweep = true;  // (never seen: four pairs a tile)
        else {
        for (uint32_t e = threadIdx.x; e < n_edges; e += kThreads) rw_union(par, edge[e] >> 16, edge[e] & 0xFFFFu);
        __syncthreads();
        RW_TICK(5)
        // (4) the footprint: the pixels of the nodes united with the seed's
        const uint32_t root = rw_find(par, s_seed_node);
        for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
            const int nc = tcomp[t];
            const uint32_t first = tfirst[t];
            uint64_t V = 0ull;
            for (int c = 0; c < nc; ++c)
                if (rw_find(par, first + (uint32_t)c) == root) V |= ((uint64_t)nhi[first + c] << 32) | nlo[first + c];
            mlo[t] = (uint32_t)V;
            mhi[t] = (uint32_t)(V >> 32);
        }
        }
        }
        if (sweep) {
            // The slow way, for logs the tables cannot take (noisy regions: many small components a tile): every tile keeps the
            // set R of its pixels reached so far; a sweep pulls in what the eight neighbours' sets touch and closes it inside the
            // tile; sweeps until nothing changes (as many as the footprint is tiles deep -- such logs are blobs, not lines).
            // R lives where the nodes' pixels were; stale or torn reads of a neighbour's R only delay (R grows monotonically).
            __syncthreads();
            for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
                const uint64_t R0 = tid[t] == seed_tile ? seed_bit : 0ull;
                nlo[t] = (uint32_t)R0;
                nhi[t] = (uint32_t)(R0 >> 32);
            }
            bool failed = true;
            for (uint32_t it = 0; it < 8u * (uint32_t)kTiles + 64u; ++it) {
                __syncthreads();
                if (threadIdx.x == 0) s_over = 0u;  // (reused: "something changed in this sweep")
                __syncthreads();
                bool changed = false;
                for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
                    const uint64_t M = ((uint64_t)mhi[t] << 32) | mlo[t];
                    if (M == 0ull) continue;
                    const uint64_t R = ((uint64_t)nhi[t] << 32) | nlo[t];
                    const uint32_t tile = tid[t];
                    uint64_t in = 0ull;
#pragma unroll
                    for (int d = 0; d < 8; ++d) {
                        // the neighbour in direction d (E, W, S, N, SE, NW, SW, NE) and what of this tile its set touches
                        const uint32_t delta = d == 0 ? 1u : d == 1 ? 0xFFFFFFFFu : d == 2 ? 0x10000u : d == 3 ? 0xFFFF0000u : d == 4 ? 0x10001u
                                               : d == 5 ? 0xFFFEFFFFu : d == 6 ? 0xFFFFu : 0xFFFF0001u;
                        const uint32_t key = tile + delta + 1u;
                        uint32_t hs = (key * 2654435761u) >> kHashShift, cur;
                        while ((cur = hkey[hs]) != key && cur != 0u) hs = (hs + 1u) & (uint32_t)(kHash - 1);
                        if (cur != key) continue;
                        const uint32_t u = hidx[hs];
                        const uint64_t a = ((uint64_t)nhi[u] << 32) | nlo[u];
                        uint64_t e;
                        if (d == 0) {  // from the east neighbour: its column 0 touches this tile's column 7
                            e = (a & kCol0) << 7;
                            e |= (e << 8) | (e >> 8);
                        } else if (d == 1) {
                            e = (a >> 7) & kCol0;
                            e |= (e << 8) | (e >> 8);
                        } else if (d == 2) {  // from the south neighbour: its row 0 touches this tile's row 7
                            e = a & 0xFFull;
                            e = ((e | (e << 1) | (e >> 1)) & 0xFFull) << 56;
                        } else if (d == 3) {
                            e = a >> 56;
                            e = (e | (e << 1) | (e >> 1)) & 0xFFull;
                        } else if (d == 4) {
                            e = (a & 1ull) << 63;
                        } else if (d == 5) {
                            e = a >> 63;
                        } else if (d == 6) {  // south-west neighbour: its (row 0, column 7) touches this tile's (row 7, column 0)
                            e = ((a >> 7) & 1ull) << 56;
                        } else {
                            e = ((a >> 56) & 1ull) << 7;
                        }
                        in |= e;
                    }
                    uint64_t x = R | (in & M), prev;
                    do {
                        prev = x;
                        x = dilate8(x) & M;
                    } while (x != prev);
                    if (x != R) {
                        nlo[t] = (uint32_t)x;
                        nhi[t] = (uint32_t)(x >> 32);
                        changed = true;
                    }
                }
                if (changed) s_over = 1u;
                __syncthreads();
                if (uni(s_over) == 0u) {
                    failed = false;
                    break;
                }
            }
            __syncthreads();
            if (failed) {  // (never: a footprint is at most as deep as it has tiles)
                if (threadIdx.x == 0) {
                    A.flags[k] = kFlagIncomplete;
                    atomicMin(&A.ctrl[kCtrlBarrier], k);
                    A.log_len[k] = 0u;
                }
                continue;
            }
            for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
                mlo[t] = nlo[t];
                mhi[t] = nhi[t];
            }
            if (threadIdx.x == 0) atomicAdd(&A.ctrl[kCtrlLogGiveUp], 1u);
        }
        __syncthreads();
        // (4) the footprint: the pixels of the nodes united with the seed's
        uint32_t my_cnt = 0u;
        for (uint32_t t = threadIdx.x; t < n; t += kThreads) {
            const uint64_t V = ((uint64_t)mhi[t] << 32) | mlo[t];
            if (V != 0ull) {  // the log shrinks with the footprint (every record was read into LDS above; their order is free)
                const uint32_t pos = atomicAdd(&s_nout, 1u);
                rec[3u * pos] = tid[t];
                rec[3u * pos + 1u] = (uint32_t)V;
                rec[3u * pos + 2u] = (uint32_t)(V >> 32);
            }
            my_cnt += (uint32_t)__popcll(V);
        }
        if (my_cnt) atomicAdd(&s_cnt, my_cnt);
        __syncthreads();
        RW_TICK(6)
        // (5) stamps, as stamp_footprint leaves them
        bool foreign = false;
        for (uint32_t i0 = (uint32_t)wave * kRwBatch; i0 < n; i0 += (uint32_t)kWaves * kRwBatch) {
            uint32_t old[kRwBatch];
#pragma unroll
            for (int j = 0; j < (int)kRwBatch; ++j) {
                old[j] = kLabelFree;
                const uint32_t i = i0 + (uint32_t)j;
                if (i < n) {
                    const uint32_t tile = tid[i];
                    const uint64_t V = ((uint64_t)mhi[i] << 32) | mlo[i];
                    if ((V >> lane) & 1ull) {
                        const size_t q = (size_t)((tile >> 16) * 8 + lr) * A.w + ((tile & 0xFFFFu) * 8 + lc);
                        old[j] = atomicMin(&A.label[q], mine);
                        A.dirty[q >> 8] = 1;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < (int)kRwBatch; ++j) {
                if (old[j] > mine) {
                    if (old[j] != kLabelFree) A.blocked[old[j] & ~kMarkBit] = 1u;
                } else if (old[j] < mine && old[j] >= kMarkBit) {
                    foreign = true;
                }
            }
        }
        if (__ballot(foreign) && lane == 0) s_blocked = 1u;
        __syncthreads();
        RW_TICK(7)
        if (threadIdx.x == 0) {
            A.count[k] = s_cnt;
            A.log_len[k] = s_nout | kLogShrunk;
            if (s_blocked) A.blocked[k] = 1u;
            A.flags[k] |= n << 8;  // (diagnostics: records looked at)
            atomicAdd(&A.ctrl[kCtrlLogWalks], 1u);
        }
    }
}

// Partial commits (round 3).  A seed that is blocked -- a lower active seed reaches some pixel of its footprint -- still
// owns, for certain, the part of the footprint that is connected to its seed pixel through pixels that carry ITS stamp
// (the lowest stamp wins a pixel, so no lower active seed reaches those) or its label from an earlier round: nothing a
// lower seed does can take them away, and the ordered flood (filter.cpp:110-153) gives them to this seed.  After the
// round's explorations one wavefront per blocked seed walks that part (the same tile walk, acceptance = "my stamp or my
// label"), turns its stamps into labels and clears their direction mask, so that
//   - seeds whose pixel lies in it are dead now, not after the round in which this seed finally comes out unblocked,
//   - the seed's later explorations start from what it owns and meet fewer foreign stamps.
// Only for seeds below the round's barrier (all lower seeds have stamped completely) whose exploration finished.  A
// walk that outgrows the first storage tier simply stops: any connected part is as safe as the whole.
__global__ __launch_bounds__(64) void flood_partial_commit_kernel(FloodArgs A, uint8_t* __restrict__ dmask_rw) {
    __shared__ uint32_t s_ring[3][kRingT];
    __shared__ uint32_t s_hash[3][kHashT];
    __shared__ uint32_t s_pend[2][kPend];
    __shared__ uint8_t s_ord[kHashT];
    const int lane = threadIdx.x & 63;
    if (giant_pending(A)) return;
    const uint32_t* __restrict__ act = act_now(A);
    const uint32_t n_act = uni(A.ctrl[kCtrlNAct]), window = uni(A.ctrl[kCtrlWindow]), barrier = uni(A.ctrl[kCtrlBarrier]);
    LdsStore L{s_ring[0], s_ring[1], s_ring[2], s_hash[0], s_hash[1], s_hash[2], s_ord};
    Pending P{s_pend[0], s_pend[1]};
    for (uint32_t ai = uni(blockIdx.x); ai < n_act; ai += gridDim.x) {
        const uint32_t k = uni(act[ai]);
        if (k >= window || k >= barrier) continue;
        const uint32_t fl = uni(A.flags[k]);
        if (uni(A.count[k]) == 0u || uni(A.blocked[k]) == 0u || (fl & (kFlagIncomplete | kFlagSelfFail))) continue;
        const int s = (int)uni((uint32_t)A.seed_idx[k]);
        const uint32_t seed_label = uni(A.label[s]);
        if (seed_label != (kMarkBit | k) && seed_label != k) continue;  // its own pixel is contested (or taken)
        for (int i = lane; i < LdsStore::kHashN; i += 64) L.hk[i] = 0u;
        P.pt[lane] = 0u;
        const int sr = s / A.w, sc = s - sr * A.w;
        L.put(0u, ((uint32_t)(sr >> 3) << 16) | (uint32_t)(sc >> 3), 1ull << ((sr & 7) * 8 + (sc & 7)));
        WalkState st{0u, 1u, 0u, 0u, false, 0u, 0u};
        (void)walk<LdsStore, 1>(A, k, 0, 0.f, 0.f, 0.f, L, P, st, lane, false, dmask_rw);
    }
}

// Does seed k commit in this round?  Its walk finished (count > 0, not incomplete), no lower active seed reaches its
// footprint (not blocked), and it lies below the lowest seed whose walk ran out of storage (barrier).  Everything this
// reads is final once the round's exploration kernels are done, so the commit pass and the survivors pass each
// evaluate it where they need it (a separate "decide" launch per round used to).
__device__ __forceinline__ bool seed_commits(const FloodArgs& A, uint32_t k, uint32_t barrier) {
    const uint32_t fl = A.flags[k];
    return A.count[k] > 0u && A.blocked[k] == 0u && !(fl & (kFlagIncomplete | kFlagSelfFail)) && k < barrier;
}

// Stamps of committed seeds become labels, all other stamps are erased.  A committed pixel also loses its direction
// mask: the walks then reject it on the mask alone and never load the label image (a quarter of their gathers).
// Only the parts of the label image that were stamped in this round are read: every stamp marks its run of 256
// consecutive pixels in A.dirty, a wavefront takes one run (four pixels per lane) and clears the mark.  After the first
// two rounds few runs are marked, and the pass costs a launch instead of a sweep over 4 bytes per pixel.
__global__ __launch_bounds__(256) void flood_commit_pixels_kernel(FloodArgs A, uint32_t* __restrict__ label, size_t npix,
                                                                  uint8_t* __restrict__ dmask) {
    const uint32_t* __restrict__ ctrl = A.ctrl;
    if (ctrl[kCtrlNAct] == 0u || ctrl[kCtrlGiantStep] != 0u) return;  // a round enqueued past the end, or behind a request for a giant step
    const uint32_t barrier = ctrl[kCtrlBarrier];
    const int lane = threadIdx.x & 63;
    const uint32_t n_runs = (uint32_t)((npix + 255) >> 8);
    for (uint32_t run = blockIdx.x * 4u + (threadIdx.x >> 6); run < n_runs; run += gridDim.x * 4u) {
        if (uni((uint32_t)A.dirty[run]) == 0u) continue;
        if (lane == 0) A.dirty[run] = 0;
        const size_t i0 = ((size_t)run << 8) + (size_t)lane * 4;
        uint32_t v[4] = {kLabelFree, kLabelFree, kLabelFree, kLabelFree};
        if (i0 + 3 < npix) {
            const uint4 q = *reinterpret_cast<const uint4*>(label + i0);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
        } else {
            for (int j = 0; j < 4; ++j)
                if (i0 + j < npix) v[j] = label[i0 + j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (v[j] >= kMarkBit && v[j] != kLabelFree) {
                const uint32_t k = v[j] & ~kMarkBit;
                const bool committed = seed_commits(A, k, barrier);
                label[i0 + j] = committed ? k : kLabelFree;
                if (committed) dmask[i0 + j] = 0;
            }
        }
    }
}

// What the device tells the host that enqueues rounds just in time (FloodBuffers::host_progress): ONE 64-bit word, so that
// every look is a consistent report -- n_left (29 bits) | stalled << 29 | giant step asked for << 30 | a flood of more than
// kHugeFlood pixels has been committed << 31 | rounds with work so far << 32 (12 bits) | giant steps done << 44 (16 bits) |
// calm << 60 (every seed has walked and none outgrew the first storage tier: footprints only shrink, so none ever will --
// the host leaves the second tier's launches out of the rounds it enqueues from then on) |
// 1 << 63 (a report: the host zeroes the word before the frame).  (Six separate words, the count of rounds stored last,
// let the host see "giant step asked for" beside a stale "no seeds left" and take the flood for finished.)
__device__ __forceinline__ void flood_report(uint32_t* host_progress, uint32_t rounds, uint32_t n_left, bool stalled, bool want_giant,
                                             uint32_t giants_done, bool huge, bool calm = false) {
    const unsigned long long w = (unsigned long long)(n_left & 0x1FFFFFFFu) | ((unsigned long long)(stalled ? 1u : 0u) << 29) |
                                 ((unsigned long long)(want_giant ? 1u : 0u) << 30) | ((unsigned long long)(huge ? 1u : 0u) << 31) |
                                 ((unsigned long long)min(rounds, 0xFFFu) << 32) | ((unsigned long long)min(giants_done, 0xFFFFu) << 44) |
                                 ((unsigned long long)(calm ? 1u : 0u) << 60) | (1ull << 63);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(host_progress), w, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// End of a round (one thread: the last workgroup of the survivors pass): the next list becomes the current one.  A round without progress (possible only
// when storage ran out on the lowest active seed) stops the rounds and leaves the rest to the ordered tail.
__device__ void flood_advance(uint32_t* __restrict__ ctrl, uint32_t win_shift, uint32_t regional_min, uint32_t hold_min_big,
                              uint32_t* host_progress, uint32_t hold_release, uint32_t giant_step, uint32_t giant_many) {
    const uint32_t n_act = ld_agent(&ctrl[kCtrlNAct]);
    if (n_act == 0u) {  // (a round enqueued past the end -- or a frame without seeds: the host must not wait for more)
        if (host_progress)
            flood_report(host_progress, max(ld_agent(&ctrl[kCtrlRounds]), 1u), 0u, false, false, ld_agent(&ctrl[kCtrlGiantDone]),
                         ld_agent(&ctrl[kCtrlMaxFlood]) != 0u);
        return;
    }
    const uint32_t n_seeds = ld_agent(&ctrl[kCtrlNSeeds]), win_hold = ld_agent(&ctrl[kCtrlWinHold]);
    const uint32_t n_next = ld_agent(&ctrl[kCtrlNNext]);
    const uint32_t window = ld_agent(&ctrl[kCtrlWindow]);
    const bool moved = ld_agent(&ctrl[kCtrlNCommit]) > 0u || n_next < n_act;
    // the lowest survivor is a marked giant: its flood is the next thing the ordered algorithm does, and the whole device
    // does it (giant step) before the next round with work
    const bool want_giant = giant_step != 0u && n_next > 0u && ld_agent(&ctrl[kCtrlGiantLowNext]) != 0xFFFFFFFFu &&
                            ld_agent(&ctrl[kCtrlGiantLowNext]) == ld_agent(&ctrl[kCtrlLowestNext]);
    const bool progress = moved || window < n_seeds || want_giant;
    // Window of the next round.  Staged start: it grows by << win_shift up to the seed count.  Hold-back: once a
    // full round has shown walks that outgrow the first storage tier (a frame with long edges or large smooth
    // regions: the weakest seeds, with the lowest thresholds, own the largest footprints and stay blocked for most
    // of the rounds, re-walking them every time), the window drops to win_hold and the weakest seeds wait until every
    // seed below it is resolved; then it opens for good.  Any prefix of the seed order is a valid window.
    bool staged = ld_agent(&ctrl[kCtrlStaged]) != 0u;
    unsigned long long grown = (unsigned long long)window << (staged ? 1u : win_shift);
    if (grown > n_seeds) grown = n_seeds;
    const uint32_t phase = ld_agent(&ctrl[kCtrlPhase]);
    if (n_next > 0u && giants_many(ctrl, giant_many)) {
        // a frame of overlapping giants (kCtrlStaged): the strongest quarter first from here on, no hold-back line besides
        ctrl[kCtrlStaged] = 1u;
        ctrl[kCtrlPhase] = 2u;
        grown = n_seeds >> 2;
    } else if (phase == 0u && window >= n_seeds && win_hold < n_seeds && ld_agent(&ctrl[kCtrlBigTotal]) >= hold_min_big && n_next > 0u) {
        grown = win_hold;
        ctrl[kCtrlPhase] = 1u;
    } else if (phase == 1u) {
        if (window < win_hold) {  // (a staged start that began below the hold line keeps growing up to it)
            if (grown > win_hold) grown = win_hold;
        } else {
            // released when (next to) nothing below the line is active any more: the last few seeds there are a chain
            // of small dependent floods, one round each, that need not keep everybody else waiting -- or when a round
            // moved nothing (storage ran out on the lowest active seed): the full window lets the next round detect
            // the stall
            grown = window;
            if (ld_agent(&ctrl[kCtrlBelow]) <= hold_release || !moved) {
                grown = n_seeds;
                ctrl[kCtrlPhase] = 2u;
            }
        }
    }
    // giants: the window closes in front of the lowest marked seed until that seed is the lowest active one
    {
        const uint32_t lowest = ld_agent(&ctrl[kCtrlLowestNext]), giant = ld_agent(&ctrl[kCtrlGiantLowNext]);
        // (a round under the rule that moved nothing -- the lowest seed's walk ran out of slabs -- is followed by one without
        // it, so that the full window can show the stall)
        const bool skip_rule = ld_agent(&ctrl[kCtrlGiantRuled]) != 0u && !moved && !want_giant;
        bool ruled = false;
        const unsigned long long grown_free = grown;
        if (giant != 0xFFFFFFFFu && !skip_rule) {
            // The window ends behind the lowest marked seed plus room for about kGiantProbes marked seeds more (their density
            // over the rest of the order taken as even): those walk again up to the team's table -- most fit by now, what the
            // seeds below them have committed is gone from their footprints -- and are ordinary seeds from then on; one that
            // still does not fit stays marked (and is the round's barrier).  Only as the lowest active seed may it go on.
            const uint32_t n_marked = max(ld_agent(&ctrl[kCtrlGiantCountNext]), 1u);
            const unsigned long long rest = n_seeds > giant ? (unsigned long long)(n_seeds - giant) : 1ull;
            const unsigned long long line = (unsigned long long)giant + 1ull + (unsigned long long)kGiantProbes * rest / n_marked;
            (void)lowest;
            if (line < grown) {
                grown = line;
                ruled = true;
            }
        }
        // ... and in front of the lowest seed that waits for its blocker (never the lowest active seed: its blocker is lower)
        const uint32_t defer = ld_agent(&ctrl[kCtrlDeferLowNext]);
        ctrl[kCtrlDeferLow] = defer;
        ctrl[kCtrlDeferLowNext] = 0xFFFFFFFFu;
        if (defer < grown && defer > lowest) grown = defer;
        ctrl[kCtrlWindowFree] = (uint32_t)grown_free;  // (the window before the two rules cut it)
        ctrl[kCtrlGiantRuled] = ruled ? 1u : 0u;
        ctrl[kCtrlLowest] = lowest;
        ctrl[kCtrlGiantLow] = giant;
        ctrl[kCtrlLowestNext] = 0xFFFFFFFFu;
        ctrl[kCtrlGiantLowNext] = 0xFFFFFFFFu;
        ctrl[kCtrlGiantCountNext] = 0u;
    }
    ctrl[kCtrlWindow] = (uint32_t)grown;
    ctrl[kCtrlBelow] = 0u;
    ctrl[kCtrlRounds] = ld_agent(&ctrl[kCtrlRounds]) + 1u;
    ctrl[kCtrlNRemain] = n_next;
    if (!progress) ctrl[kCtrlStall] = 1u;
    ctrl[kCtrlNAct] = progress ? n_next : 0u;
    ctrl[kCtrlNNext] = 0u;
    ctrl[kCtrlNCommit] = 0u;
    ctrl[kCtrlBarrier] = ld_agent(&ctrl[kCtrlBarrierNext]);  // (0xFFFFFFFF unless the next round cannot reach its whole list)
    ctrl[kCtrlBarrierNext] = 0xFFFFFFFFu;
    ctrl[kCtrlSlabs] = 0u;
    ctrl[kCtrlNBig] = 0u;
    ctrl[kCtrlNMulti] = progress ? ld_agent(&ctrl[kCtrlNMultiNext]) : 0u;
    ctrl[kCtrlNMultiNext] = 0u;
    ctrl[kCtrlBigSeen] = ld_agent(&ctrl[kCtrlBigLong]) >= regional_min ? 1u : 0u;
    ctrl[kCtrlGiantStep] = (progress && want_giant) ? ld_agent(&ctrl[kCtrlLowest]) + 1u : 0u;  // (kCtrlLowest: the lowest survivor, set above)
    ctrl[kCtrlGiantReuse] = 0u;  // (a round has run: the masks of the last step are history)
    if (host_progress)  // the host enqueues the next round when it sees this one over and seeds left (flood_enqueue)
        flood_report(host_progress, ld_agent(&ctrl[kCtrlRounds]), progress ? n_next : 0u, !progress, progress && want_giant,
                     ld_agent(&ctrl[kCtrlGiantDone]), ld_agent(&ctrl[kCtrlMaxFlood]) != 0u,
                     window >= n_seeds && ld_agent(&ctrl[kCtrlBigTotal]) == 0u && ld_agent(&ctrl[kCtrlSlabTotal]) == 0u &&
                         ld_agent(&ctrl[kCtrlQuietMiss]) == 0u);
}

// (Commit pass and survivors pass in ONE launch -- blocked marks in two alternating buffers, counts and flags written
// by every exploration instead of reset here, the survivors reading stamps as they find them, the workgroups counting
// themselves off on a tree of counters -- was built and is exact, but saves nothing: 30, 23, 17, 13, 11, 10 us per
// round against 16 + 13, 14 + 10, 9 + 9, ... for the two launches, single frames and batches unchanged.  With one
// __threadfence per workgroup of the 4336 it took 65-140 us per round: the fence, not the atomics.)
// After the commit: which seeds go on to the next round?
__global__ __launch_bounds__(256) void flood_survivors_kernel(FloodArgs A, uint8_t* __restrict__ state,
                                                              int32_t* __restrict__ seed_size) {
    if (giant_pending(A)) return;  // (nothing of this round ran: not counted, nothing reported)
    const uint32_t* __restrict__ act = act_now(A);
    uint32_t* __restrict__ act_next = act_other(A);
    const uint32_t n_act = A.ctrl[kCtrlNAct];
    const uint32_t n_pad = (n_act + 255u) & ~255u;  // whole workgroups take part in the ballots and barriers
    const uint32_t window = A.ctrl[kCtrlWindow];
    const uint32_t barrier = A.ctrl[kCtrlBarrier];
    // Counters of the control block are added to once per workgroup: an atomic per wavefront was 640 atomics on one
    // address in the first round, which the L2 executes one after the other (28 us for 40 000 seeds; 10 us now).
    __shared__ uint32_t s_cnt[4][5];  // per wavefront: survivors, committed, survivors below the window, pixels and steps walked
    __shared__ uint32_t s_base;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t ai = blockIdx.x * 256 + threadIdx.x; ai < n_pad; ai += gridDim.x * 256) {
        bool a = false, done = false;
        uint32_t k = 0, wpx = 0, wst = 0, dfr = 0xFFFFFFFFu;
        if (ai < n_act) {
            k = act[ai];
            wpx = A.count[k];        // what this round's exploration of k walked (0 if it did not walk)
            wst = A.flags[k] >> 8;
            if (A.flags[k] & kFlagSelfFail) {  // flood() accepted nothing, not even the seed
                state[k] = 2;
                seed_size[k] = 0;
            } else if (seed_commits(A, k, barrier)) {  // the commit pass has turned its stamps into labels
                state[k] = 2;
                seed_size[k] = (int32_t)A.count[k];
                done = true;
                if (A.count[k] > kHugeFlood) atomicMax(&A.ctrl[kCtrlMaxFlood], A.count[k]);  // (rare)
            } else if (state[k] == 0) {
                const uint32_t own_label = A.label[A.seed_idx[k]];
                if (own_label < kMarkBit && own_label != k) {  // its pixel now belongs to another seed's flood: skipped forever
                    state[k] = 2;
                    seed_size[k] = 0;
                } else {
                    a = true;
                    A.blocked[k] = 0u;
                    A.count[k] = 0u;
                    A.flags[k] = 0u;
                    // A survivor that left way-points on a long footprint: the coming round walks it from all of them at once
                    // (team_walk, kMulti) in a launch of its own BESIDE the round's exploration, which passes it over (tier
                    // bit 1).  (A few hundred such seeds a round at most: one atomic each.)
                    uint8_t t = (uint8_t)(A.tier[k] & 5u);  // (bit 0: outgrew the first tier; bit 2: a giant, held back)
                    if (A.multi_next != 0u && k < window && k < A.wp_cap) {
                        const uint32_t hdr = A.waypoints[(size_t)k * kFloodWpWords];
                        if (hdr != 0u && (hdr >> 8) <= kWpMaxTiles) {
                            const uint32_t pos = atomicAdd(&A.ctrl[kCtrlNMultiNext], 1u);
                            if (pos < kBigCap) {
                                A.multi_list[pos] = k;
                                t |= 2u;
                            }
                        }
                    }
                    A.tier[k] = t;
                    // its last long walk was blocked by a lower seed that is still unresolved (see flood_explore_team_kernel):
                    // it waits.  (state[] of the blocker may be written in this very pass: read as unresolved, the seed waits a
                    // round longer.)
                    if (A.blk != nullptr) {
                        const uint32_t bk = A.blk[k];
                        if (bk != 0xFFFFFFFFu) {
                            if (state[bk] != 0) A.blk[k] = 0xFFFFFFFFu;
                            else dfr = k;
                        }
                    }
                }
            }
        }
        // next round's active list: the order of the list does not matter
        const uint64_t m = __ballot(a), md = __ballot(done), mb = __ballot(a && k < window);
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            wpx += (uint32_t)__shfl_xor((int)wpx, off);
            wst += (uint32_t)__shfl_xor((int)wst, off);
        }
        // lowest survivor, and lowest survivor that is marked as a giant (the coming round's window: flood_advance)
        uint32_t kmin = a ? k : 0xFFFFFFFFu, gmin = (a && (A.tier[k] & 4u)) ? k : 0xFFFFFFFFu;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
            gmin = min(gmin, (uint32_t)__shfl_xor((int)gmin, off));
        }
        const uint64_t mg = __ballot(a && (A.tier[k] & 4u));
        if (__ballot(dfr != 0xFFFFFFFFu) != 0ull) {  // (rare)
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) dfr = min(dfr, (uint32_t)__shfl_xor((int)dfr, off));
            if (lane == 0) atomicMin(&A.ctrl[kCtrlDeferLowNext], dfr);
        }
        if (lane == 0) {
            if (kmin != 0xFFFFFFFFu) atomicMin(&A.ctrl[kCtrlLowestNext], kmin);
            if (gmin != 0xFFFFFFFFu) {
                atomicMin(&A.ctrl[kCtrlGiantLowNext], gmin);
                atomicAdd(&A.ctrl[kCtrlGiantCountNext], (uint32_t)__popcll(mg));
            }
        }
        if (lane == 0) {
            s_cnt[wave][0] = (uint32_t)__popcll(m);
            s_cnt[wave][1] = (uint32_t)__popcll(md);
            s_cnt[wave][2] = (uint32_t)__popcll(mb);
            s_cnt[wave][3] = wpx;
            s_cnt[wave][4] = wst;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            const uint32_t na = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
            const uint32_t nd = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
            const uint32_t nb = s_cnt[0][2] + s_cnt[1][2] + s_cnt[2][2] + s_cnt[3][2];
            s_base = na ? atomicAdd(&A.ctrl[kCtrlNNext], na) : 0u;
            if (nd) atomicAdd(&A.ctrl[kCtrlNCommit], nd);
            if (nb) atomicAdd(&A.ctrl[kCtrlBelow], nb);
            const uint32_t np_ = s_cnt[0][3] + s_cnt[1][3] + s_cnt[2][3] + s_cnt[3][3];
            const uint32_t ns_ = s_cnt[0][4] + s_cnt[1][4] + s_cnt[2][4] + s_cnt[3][4];
            if (np_) atomicAdd(reinterpret_cast<unsigned long long*>(&A.ctrl[kCtrlWalked]), (unsigned long long)np_);
            if (ns_) atomicAdd(reinterpret_cast<unsigned long long*>(&A.ctrl[kCtrlSteps]), (unsigned long long)ns_);
        }
        __syncthreads();
        if (a) {
            uint32_t base = s_base;
            for (int w2 = 0; w2 < wave; ++w2) base += s_cnt[w2][0];
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            act_next[pos] = k;
            if (pos >= A.next_reach) atomicMin(&A.ctrl[kCtrlBarrierNext], k);  // (never, in practice: see enqueue_round)
        }
        __syncthreads();  // s_cnt and s_base are rewritten by the next pass of the loop
    }
    // the workgroup that finishes last closes the round (every other one has read the control block and added its
    // counts by then)
    __shared__ uint32_t s_closed;
    __syncthreads();
    if (threadIdx.x == 0) {
        s_closed = 0u;
        __threadfence();
        if (atomicAdd(&A.ctrl[kCtrlDone], 1u) == gridDim.x - 1u) {
            __threadfence();
            A.ctrl[kCtrlDone] = 0u;
            flood_advance(A.ctrl, A.win_shift, A.t1_regional_min, A.hold_min_big, A.host_progress, A.hold_release, A.giant_step, A.giant_many);
            __threadfence();
            s_closed = 1u;
        }
    }
    // The round that leaves no seeds (or stalls) hands the host the control block as well: the host that watched the rounds
    // (flood_enqueue) then enqueues no copy of it -- a launch and 7 us of an idle stream between the flood and the fit.
    if (A.host_ctrl == nullptr) return;
    __syncthreads();
    if (s_closed != 0u && threadIdx.x < kCtrlWords && ld_agent(&A.ctrl[kCtrlNAct]) == 0u)
        __hip_atomic_store(&A.host_ctrl[threadIdx.x], ld_agent(&A.ctrl[threadIdx.x]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void flood_init_seeds_kernel(const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                               uint32_t* __restrict__ act, uint8_t* __restrict__ state,
                                                               uint8_t* __restrict__ tier, uint32_t* __restrict__ blocked,
                                                               uint32_t* __restrict__ count, uint32_t* __restrict__ flags,
                                                               int32_t* __restrict__ seed_size, uint32_t* __restrict__ ctrl,
                                                               uint8_t* __restrict__ dirty, uint32_t n_runs,
                                                               int win_first_shift, int hold_pct, uint32_t hold_from_start,
                                                               uint32_t* __restrict__ label, size_t npix,
                                                               uint32_t* __restrict__ waypoints, uint32_t wp_cap,
                                                               uint32_t* __restrict__ log_len, uint32_t log_seeds, uint32_t dense_div,
                                                               uint32_t staged_from_start, uint32_t* __restrict__ blk) {
    // (the label image is set to "free" here as well: one launch less in front of the first round)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < npix; i += (size_t)gridDim.x * 256) label[i] = kLabelFree;
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n_seeds = min(*n_ptr, cap);
    if (k == 0u) {
        // staged start and hold-back line (see kCtrlWindow, flood_advance)
        uint32_t win_first = win_first_shift > 0 ? max(1024u, n_seeds >> win_first_shift) : n_seeds;
        // A frame where every twelfth pixel or more is a seed is one of exact ties (a periodic pattern without noise: every
        // pixel of a flank a seed of the same magnitude, hundreds of seeds with ONE footprint): the strongest eighth walks
        // first, commits the flanks, and the others die unwalked (stripes of period 6 at 1080p: 687 564 seeds, 639 components).
        if (dense_div != 0u && win_first_shift <= 0 && n_seeds > (uint32_t)(npix / dense_div)) win_first = max(1024u, n_seeds >> 3);
        if (win_first > n_seeds) win_first = n_seeds;
        uint32_t win_hold = (hold_pct > 0 && hold_pct < 100) ? (uint32_t)((unsigned long long)n_seeds * (uint32_t)hold_pct / 100u) : n_seeds;
        if (win_hold < 1024u) win_hold = n_seeds;  // not worth another phase
        const bool hold_now = hold_from_start != 0u && win_hold < n_seeds;
        if (hold_now && win_first > win_hold) win_first = win_hold;
        ctrl[kCtrlNSeeds] = n_seeds;
        ctrl[kCtrlWinHold] = win_hold;
        ctrl[kCtrlWindow] = win_first;
        ctrl[kCtrlDone] = 0u;
        ctrl[kCtrlBelow] = 0u;
        ctrl[kCtrlNBig] = 0u;
        ctrl[kCtrlBigTotal] = 0u;
        ctrl[kCtrlBigLong] = 0u;
        ctrl[kCtrlBigSeen] = 0u;
        ctrl[kCtrlMulti] = 0u;
        ctrl[kCtrlNMulti] = 0u;
        ctrl[kCtrlNMultiNext] = 0u;
        ctrl[kCtrlLowest] = 0u;
        ctrl[kCtrlLowestNext] = 0xFFFFFFFFu;
        ctrl[kCtrlGiantLow] = 0xFFFFFFFFu;
        ctrl[kCtrlGiantLowNext] = 0xFFFFFFFFu;
        ctrl[kCtrlGiantRuled] = 0u;
        ctrl[kCtrlGiantCountNext] = 0u;
        ctrl[kCtrlGiants] = 0u;
        ctrl[kCtrlGiantStep] = 0u;
        ctrl[kCtrlGiantReuse] = 0u;
        ctrl[kCtrlDeferLow] = 0xFFFFFFFFu;
        ctrl[kCtrlDeferLowNext] = 0xFFFFFFFFu;
        ctrl[kCtrlStaged] = staged_from_start;  // (the context's last frame was one of overlapping giants: FloodBuffers::staged_from_start)
        ctrl[kCtrlMaxFlood] = 0u;
        ctrl[kCtrlGiantDone] = 0u;
        ctrl[kCtrlGiantPx] = 0u;
        ctrl[kCtrlGiantBlocks] = 0u;
        ctrl[kCtrlLogTotal] = 0u;
        ctrl[kCtrlLogWalks] = 0u;
        ctrl[kCtrlLogGiveUp] = 0u;
        ctrl[kCtrlSlabTotal] = 0u;
        ctrl[kCtrlQuietMiss] = 0u;
        ctrl[kCtrlTeamDone] = 0u;
        ctrl[kCtrlTeamGiants] = 0u;
        ctrl[kCtrlBarrier] = 0xFFFFFFFFu;
        ctrl[kCtrlBarrierNext] = 0xFFFFFFFFu;
        ctrl[kCtrlSlabs] = 0u;
        ctrl[kCtrlNAct] = n_seeds;
        ctrl[kCtrlNCommit] = 0u;
        ctrl[kCtrlNNext] = 0u;
        ctrl[kCtrlPhase] = hold_now ? 1u : 0u;
        ctrl[kCtrlRounds] = 0u;
        ctrl[kCtrlStall] = 0u;
        ctrl[kCtrlNRemain] = n_seeds;
        ctrl[kCtrlWalked] = ctrl[kCtrlWalked + 1] = 0u;
        ctrl[kCtrlSteps] = ctrl[kCtrlSteps + 1] = 0u;
    }
    for (uint32_t r = k; r < n_runs; r += gridDim.x * 256) dirty[r] = 0;  // (all clear after a flood that ran to its end)
    if (k >= n_seeds) return;
    if (k < wp_cap) waypoints[(size_t)k * kFloodWpWords] = 0u;
    if (k < log_seeds) log_len[k] = 0u;
    act[k] = k;
    state[k] = 0;
    tier[k] = 0;
    if (blk) blk[k] = 0xFFFFFFFFu;
    blocked[k] = 0u;
    count[k] = 0u;
    flags[k] = 0u;
    seed_size[k] = 0;
}

// ---- The giant step: the lowest active seed's flood by the whole device -----------------------------------------------------
// (see kCtrlGiantStep.)  The lowest active seed is never blocked: what it reaches is what the ordered flood (filter.cpp:110-153
// under line_detector.cpp:98-119) gives it, the connected component around its pixel of {acceptable for this seed, not
// committed} -- and nothing about that set is speculative.  A team of eight wavefronts used to walk it tile after tile
// through a global slab (a region of 140 000 pixels: 11.7 ms of a 1080p frame; a ring of a noiseless radial gradient: 3-9 ms,
// sixteen of them).  Here the component is LABELLED instead of walked, by as many workgroups as the frame has tiles:
//   giant_mask_kernel    the seed's acceptance test on every pixel, as one 64-bit mask per 8x8 tile (a wavefront takes a strip
//                        of eight tiles: 256-byte rows); the in-tile components of each mask (8-neighbour closure on the mask)
//                        become the nodes of a union-find, a node's name the index of its first pixel;
//   giant_merge_kernel   a thread a tile: components of neighbouring tiles that touch across the border (east, south,
//                        south-east, south-west) are united -- compare-and-swap on roots, path halving, every access at L2;
//   giant_commit_kernel  the components united with the seed's become its flood: label = seed, direction mask cleared (the
//                        form a commit leaves), pixels counted;
//   giant_finish_kernel  the seed is retired (tier bit 3: explorations pass it over; count, blocked and flags as a finished,
//                        unblocked walk leaves them: the next round's survivors pass books it as committed), the lowest
//                        seed of the list that is still alive becomes kCtrlLowest, the giants' line is drawn again -- and
//                        if that seed is a marked giant too, the next step is asked for at once (a ring of equal seeds
//                        after another: no round in between).
// The cost does not depend on the component's shape or size: 9 bytes a pixel read once, a word a tile, a handful of atomics
// a tile border.  Exact because the step does what the reference's loop does next: every lower seed is resolved.
__device__ __forceinline__ uint64_t giant_closure(uint64_t c, uint64_t M) {
    uint64_t prev;
    do {
        prev = c;
        c = dilate8(c) & M;
        c = dilate8(c) & M;
    } while (c != prev);
    return c;
}
__device__ __forceinline__ uint32_t gu_find(uint32_t* par, uint32_t x) {
    for (;;) {
        const uint32_t p = ld_agent(&par[x]);
        if (p == x) return x;
        const uint32_t g = ld_agent(&par[p]);
        if (g != p) __hip_atomic_store(&par[x], g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (any ancestor is a valid parent)
        x = g;
    }
}
__device__ __forceinline__ void gu_union(uint32_t* par, uint32_t a, uint32_t b) {
    for (;;) {
        uint32_t ra = gu_find(par, a), rb = gu_find(par, b);
        if (ra == rb) return;
        if (rw_order(ra) < rw_order(rb)) {  // (a scrambled order of the names: see rw_order)
            const uint32_t t = ra;
            ra = rb;
            rb = t;
        }
        if (atomicCAS(&par[ra], ra, rb) == ra) return;  // (a root only ever changes through this exchange)
    }
}
__device__ __forceinline__ uint32_t giant_pixel(uint32_t tx, uint32_t ty, uint32_t bit, uint32_t w) {
    return (ty * 8u + (bit >> 3)) * w + tx * 8u + (bit & 7u);
}

__global__ __launch_bounds__(256) void giant_mask_kernel(FloodArgs A, BinTrig trig) {
    const uint32_t gs = uni(A.ctrl[kCtrlGiantStep]);
    if (gs == 0u || uni(A.ctrl[kCtrlGiantReuse]) != 0u) return;  // (reuse: see giant_finish_kernel)
    const uint32_t g = gs - 1u;
    const int lane = threadIdx.x & 63;
    const int s = (int)uni((uint32_t)A.seed_idx[g]);
    const int b = (int)uni((uint32_t)A.seed_bin[g]);
    const float thr = __uint_as_float(uni(__float_as_uint(A.seed_thr[g])));
    const float sn = trig.st[b], cs = trig.ct[b];
    const bool own = uni(A.label[s]) == g;  // it has committed a part of its flood in the rounds: its own ground
    const uint32_t bin_bit = 1u << b;
    const uint32_t tiles_x = (uint32_t)A.tiles_x, tiles_y = (uint32_t)(A.h + 7) >> 3;
    const uint32_t strips_x = (tiles_x + 7u) >> 3, n_strips = strips_x * tiles_y;
    const uint32_t uw = (uint32_t)A.w;
    for (uint32_t si = blockIdx.x * 4u + (threadIdx.x >> 6); si < n_strips; si += gridDim.x * 4u) {
        const uint32_t sy = uni(si / strips_x), sx = uni(si - sy * strips_x);
        const uint32_t x = sx * 64u + (uint32_t)lane;
        uint32_t dm[8], lab[8];
        float gx[8], gy[8];
        bool in[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const uint32_t y = sy * 8u + (uint32_t)r;
            in[r] = x < uw && y < (uint32_t)A.h;
            const uint32_t q = in[r] ? y * uw + x : 0u;
            dm[r] = ld8(A.dmask, q);
            gx[r] = ldf(A.dx, q);
            gy[r] = ldf(A.dy, q);
            lab[r] = own ? ld32(A.label, q) : kLabelFree;
        }
        uint64_t rows[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            bool ok = in[r] && (dm[r] & bin_bit) != 0u && directional(gx[r], gy[r], sn, cs) > thr;
            if (own) ok = ok || (in[r] && lab[r] == g);
            rows[r] = __ballot(ok);
        }
        // lanes 0..7: a tile each -- its mask out of the eight row ballots, its components' first pixels into the union-find
        if (lane < 8) {
            const uint32_t tx = sx * 8u + (uint32_t)lane;
            if (tx < tiles_x) {
                uint64_t M = 0ull;
#pragma unroll
                for (int r = 0; r < 8; ++r) M |= ((rows[r] >> (8 * lane)) & 0xFFull) << (8 * r);
                A.giant_mask[(size_t)sy * tiles_x + tx] = M;
                uint64_t rem = M;
                while (rem != 0ull) {
                    const uint64_t c = giant_closure(rem & (~rem + 1ull), M);
                    rem &= ~c;
                    const uint32_t p = giant_pixel(tx, sy, (uint32_t)__builtin_ctzll(c), uw);
                    A.giant_parent[p] = p;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void giant_merge_kernel(FloodArgs A) {
    if (A.ctrl[kCtrlGiantStep] == 0u || A.ctrl[kCtrlGiantReuse] != 0u) return;
    const uint32_t tiles_x = (uint32_t)A.tiles_x, tiles_y = (uint32_t)(A.h + 7) >> 3, n_tiles = tiles_x * tiles_y;
    const uint32_t uw = (uint32_t)A.w;
    uint32_t* par = A.giant_parent;
    for (uint32_t t = blockIdx.x * 256u + threadIdx.x; t < n_tiles; t += gridDim.x * 256u) {
        const uint64_t M = A.giant_mask[t];
        if (M == 0ull) continue;
        const uint32_t ty = t / tiles_x, tx = t - ty * tiles_x;
        const bool east = tx + 1u < tiles_x, south = ty + 1u < tiles_y, west = tx > 0u;
        const uint64_t Me = east ? A.giant_mask[t + 1u] : 0ull;
        const uint64_t Ms = south ? A.giant_mask[t + tiles_x] : 0ull;
        const uint64_t Mse = (south && east) ? A.giant_mask[t + tiles_x + 1u] : 0ull;
        const uint64_t Msw = (south && west) ? A.giant_mask[t + tiles_x - 1u] : 0ull;
        if (((M & kCol7) == 0ull || (Me & kCol0) == 0ull) && ((M >> 56) == 0ull || (Ms & 0xFFull) == 0ull) &&
            ((M >> 63) == 0ull || (Mse & 1ull) == 0ull) && (((M >> 56) & 1ull) == 0ull || ((Msw >> 7) & 1ull) == 0ull))
            continue;  // nothing of this tile touches a tile to the east or south of it
        uint64_t rem = M;
        while (rem != 0ull) {
            const uint64_t c = giant_closure(rem & (~rem + 1ull), M);
            rem &= ~c;
            const uint32_t pc = giant_pixel(tx, ty, (uint32_t)__builtin_ctzll(c), uw);
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                uint64_t e, Mu;
                uint32_t ux, uy;
                if (d == 0) {  // east: this component's column 7 against the neighbour's column 0
                    e = (c >> 7) & kCol0;
                    e |= (e << 8) | (e >> 8);
                    Mu = Me, ux = tx + 1u, uy = ty;
                } else if (d == 1) {  // south: its row 7 against the neighbour's row 0
                    e = c >> 56;
                    e = (e | (e << 1) | (e >> 1)) & 0xFFull;
                    Mu = Ms, ux = tx, uy = ty + 1u;
                } else if (d == 2) {  // south-east: corner against corner
                    e = c >> 63;
                    Mu = Mse, ux = tx + 1u, uy = ty + 1u;
                } else {  // south-west
                    e = ((c >> 56) & 1ull) << 7;
                    Mu = Msw, ux = tx - 1u, uy = ty + 1u;
                }
                e &= Mu;
                while (e != 0ull) {
                    const uint64_t cu = giant_closure(e & (~e + 1ull), Mu);
                    e &= ~cu;
                    gu_union(par, pc, giant_pixel(ux, uy, (uint32_t)__builtin_ctzll(cu), uw));
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void giant_commit_kernel(FloodArgs A, uint8_t* __restrict__ dmask_rw) {
    const uint32_t gs = uni(A.ctrl[kCtrlGiantStep]);
    if (gs == 0u) return;
    const uint32_t g = gs - 1u;
    __shared__ uint32_t s_root;
    const uint32_t tiles_x = (uint32_t)A.tiles_x, tiles_y = (uint32_t)(A.h + 7) >> 3;
    const uint32_t uw = (uint32_t)A.w;
    uint32_t* par = A.giant_parent;
    if (threadIdx.x == 0) {
        const uint32_t s = (uint32_t)A.seed_idx[g];
        const uint32_t sr = s / uw, sc = s - sr * uw;
        const uint64_t M = A.giant_mask[(size_t)(sr >> 3) * tiles_x + (sc >> 3)];
        const uint64_t bit = 1ull << ((sr & 7u) * 8u + (sc & 7u));
        uint32_t root = 0xFFFFFFFFu;  // (the seed's own pixel is not acceptable: flood() accepts nothing)
        if (M & bit) root = gu_find(par, giant_pixel(sc >> 3, sr >> 3, (uint32_t)__builtin_ctzll(giant_closure(bit, M)), uw));
        s_root = root;
    }
    __syncthreads();
    const uint32_t root = s_root;
    if (root == 0xFFFFFFFFu) return;
    const int lane = threadIdx.x & 63;
    // a wavefront takes 64 consecutive tiles of a tile row, a lane a tile; then the flood's pixels of those tiles are written
    // eight tiles (64 pixels a row) at a time
    const uint32_t runs_x = (tiles_x + 63u) >> 6, n_runs = runs_x * tiles_y;
    uint32_t cnt = 0u;
    for (uint32_t ri = blockIdx.x * 4u + (threadIdx.x >> 6); ri < n_runs; ri += gridDim.x * 4u) {
        const uint32_t ty = uni(ri / runs_x), rx = uni(ri - ty * runs_x);
        const uint32_t tx = rx * 64u + (uint32_t)lane;
        uint64_t F = 0ull;
        if (tx < tiles_x) {
            const uint64_t M = A.giant_mask[(size_t)ty * tiles_x + tx];
            uint64_t rem = M;
            while (rem != 0ull) {
                const uint64_t c = giant_closure(rem & (~rem + 1ull), M);
                rem &= ~c;
                if (gu_find(par, giant_pixel(tx, ty, (uint32_t)__builtin_ctzll(c), uw)) == root) F |= c;
            }
        }
        cnt += (uint32_t)__popcll(F);
        const uint64_t any = __ballot(F != 0ull);
        if (any == 0ull) continue;
        for (int sub = 0; sub < 8; ++sub) {
            if (((any >> (8 * sub)) & 0xFFull) == 0ull) continue;
            const int src = sub * 8 + (lane >> 3);
            const uint64_t Fs = ((uint64_t)(uint32_t)__shfl((int)(uint32_t)(F >> 32), src) << 32) | (uint32_t)__shfl((int)(uint32_t)F, src);
            const uint32_t x = rx * 512u + (uint32_t)sub * 64u + (uint32_t)lane;
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if ((Fs >> (8 * r + (lane & 7))) & 1ull) {
                    const uint32_t q = (ty * 8u + (uint32_t)r) * uw + x;
                    A.label[q] = g;
                    dmask_rw[q] = 0;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off);
    if (lane == 0 && cnt != 0u) atomicAdd(&A.ctrl[kCtrlGiantPx], cnt);
}

__global__ __launch_bounds__(256) void giant_finish_kernel(FloodArgs A) {
    const uint32_t gs = A.ctrl[kCtrlGiantStep];
    if (gs == 0u) return;
    const uint32_t g = gs - 1u;
    const uint32_t* __restrict__ act = act_now(A);
    const uint32_t n_act = A.ctrl[kCtrlNAct];
    const int lane = threadIdx.x & 63;
    // the lowest seed of the list that is still alive, and the lowest marked one (what flood_advance takes from the survivors
    // pass); one set of atomics a workgroup (one a wavefront was 3 700 atomics on three addresses at 1080p: 50 us)
    uint32_t kmin = 0xFFFFFFFFu, gmin = 0xFFFFFFFFu, nmark = 0u;
    for (uint32_t ai = blockIdx.x * 256u + threadIdx.x; ai < n_act; ai += gridDim.x * 256u) {
        const uint32_t k = act[ai];
        const uint8_t t = A.tier[k];
        if (k != g && (t & 8u) == 0u) {
            const uint32_t own_label = A.label[A.seed_idx[k]];
            if (own_label >= kMarkBit || own_label == k) {  // (between rounds: free, or its own ground)
                kmin = min(kmin, k);
                if (t & 4u) {
                    gmin = min(gmin, k);
                    nmark += 1u;
                }
            }
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        kmin = min(kmin, (uint32_t)__shfl_xor((int)kmin, off));
        gmin = min(gmin, (uint32_t)__shfl_xor((int)gmin, off));
        nmark += (uint32_t)__shfl_xor((int)nmark, off);
    }
    __shared__ uint32_t s_k[4], s_g[4], s_n[4];
    if (lane == 0) {
        s_k[threadIdx.x >> 6] = kmin;
        s_g[threadIdx.x >> 6] = gmin;
        s_n[threadIdx.x >> 6] = nmark;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        kmin = min(min(s_k[0], s_k[1]), min(s_k[2], s_k[3]));
        gmin = min(min(s_g[0], s_g[1]), min(s_g[2], s_g[3]));
        nmark = s_n[0] + s_n[1] + s_n[2] + s_n[3];
        if (kmin != 0xFFFFFFFFu) atomicMin(&A.ctrl[kCtrlLowestNext], kmin);
        if (gmin != 0xFFFFFFFFu) {
            atomicMin(&A.ctrl[kCtrlGiantLowNext], gmin);
            atomicAdd(&A.ctrl[kCtrlGiantCountNext], nmark);
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    __threadfence();
    if (atomicAdd(&A.ctrl[kCtrlGiantBlocks], 1u) != gridDim.x - 1u) return;
    __threadfence();
    uint32_t* ctrl = A.ctrl;
    const uint32_t total = ld_agent(&ctrl[kCtrlGiantPx]);
    // the seed leaves the rounds as a finished, unblocked walk: the coming round's survivors pass books it as committed
    A.count[g] = total;
    A.blocked[g] = 0u;
    A.flags[g] = total != 0u ? 0u : kFlagSelfFail;
    A.tier[g] = (uint8_t)((A.tier[g] & ~4u) | 8u);
    const uint32_t lowest = ld_agent(&ctrl[kCtrlLowestNext]), giant = ld_agent(&ctrl[kCtrlGiantLowNext]);
    const uint32_t n_seeds = ld_agent(&ctrl[kCtrlNSeeds]);
    // the coming round's window: the giants' rule again, with the lowest marked seed that is left (flood_advance)
    unsigned long long grown = ld_agent(&ctrl[kCtrlWindowFree]);
    bool ruled = false;
    if (giant != 0xFFFFFFFFu) {
        const uint32_t n_marked = max(ld_agent(&ctrl[kCtrlGiantCountNext]), 1u);
        const unsigned long long rest = n_seeds > giant ? (unsigned long long)(n_seeds - giant) : 1ull;
        const unsigned long long line = (unsigned long long)giant + 1ull + (unsigned long long)kGiantProbes * rest / n_marked;
        if (line < grown) {
            grown = line;
            ruled = true;
        }
    }
    {   // (and the line in front of a seed that waits for its blocker, as the last round's end drew it)
        const uint32_t defer = ld_agent(&ctrl[kCtrlDeferLow]);
        if (defer < grown && defer > lowest) grown = defer;
    }
    ctrl[kCtrlWindow] = (uint32_t)grown;
    ctrl[kCtrlGiantRuled] = ruled ? 1u : 0u;
    ctrl[kCtrlLowest] = lowest;
    ctrl[kCtrlGiantLow] = giant;
    ctrl[kCtrlLowestNext] = 0xFFFFFFFFu;
    ctrl[kCtrlGiantLowNext] = 0xFFFFFFFFu;
    ctrl[kCtrlGiantCountNext] = 0u;
    ctrl[kCtrlBarrier] = 0xFFFFFFFFu;  // (the round that follows is enqueued with its list's length known: it reaches every entry)
    ctrl[kCtrlGiantPx] = 0u;
    ctrl[kCtrlGiantBlocks] = 0u;
    const uint32_t done = ld_agent(&ctrl[kCtrlGiantDone]) + 1u;
    ctrl[kCtrlGiantDone] = done;
    const bool again = giant != 0xFFFFFFFFu && giant == lowest;
    ctrl[kCtrlGiantStep] = again ? giant + 1u : 0u;
    // The next step's seed passes the very same test as this one's (same bin, same threshold to the bit) and has committed
    // nothing of its own yet: its flood is a component of the SAME mask -- this step's commit has only taken one component
    // away -- so the step that follows leaves its mask and union-find launches at once and goes straight to the commit
    // (noiseless stripes: a hundred steps of 110 us of union-find over all the stripes, for one stripe each).
    ctrl[kCtrlGiantReuse] = (again && A.seed_bin[giant] == A.seed_bin[g] && __float_as_uint(A.seed_thr[giant]) == __float_as_uint(A.seed_thr[g]) &&
                             A.label[A.seed_idx[giant]] != giant && total != 0u)
                                ? 1u
                                : 0u;
    if (A.host_progress)
        flood_report(A.host_progress, ld_agent(&ctrl[kCtrlRounds]), n_act, false, again, done, ld_agent(&ctrl[kCtrlMaxFlood]) != 0u || total > kHugeFlood);
}

// Ordered tail: the reference's loop over an (ascending) list of remaining seeds, starting from
// the labels committed so far.  Same walk as flood_ordered_kernel.  A seed that has committed a part of its flood in the
// rounds (its pixel carries its own index) goes on from there: the walk passes through its own pixels, which it tags on
// the way (bit 30 of the label: seed indices stay below 2^29) so that each is visited once, and removes the tags at the
// end; the size it reports covers the whole flood.
constexpr uint32_t kTailTag = 0x40000000u;
__global__ __launch_bounds__(64) void flood_ordered_tail_kernel(const float* __restrict__ dx, const float* __restrict__ dy,
                                                                const uint8_t* __restrict__ dmask, int w,
                                                                const int32_t* __restrict__ seed_idx,
                                                                const int32_t* __restrict__ seed_bin,
                                                                const float* __restrict__ seed_thr,
                                                                const uint32_t* __restrict__ act, uint32_t n_act,
                                                                BinTrig trig, uint32_t* label, int32_t* seed_size,
                                                                int32_t* queue) {
    const int lane = threadIdx.x;
    const int si = lane >> 3, ni = lane & 7;
    const int dr = (ni == 2 || ni == 6 || ni == 7) ? 1 : ((ni == 3 || ni == 4 || ni == 5) ? -1 : 0);
    const int dc = (ni == 0 || ni == 4 || ni == 6) ? -1 : ((ni == 1 || ni == 5 || ni == 7) ? 1 : 0);
    const int noff = dr * w + dc;
    for (uint32_t ai = 0; ai < n_act; ++ai) {
        const uint32_t k = act[ai];
        const int sidx = seed_idx[k];
        int size = 0;
        const uint32_t seed_label = ld_agent(&label[sidx]);
        const bool own = seed_label == k;
        if (seed_label == kLabelFree || own) {
            const int b = seed_bin[k];
            const float thr = seed_thr[k];
            const float s = trig.st[b], c = trig.ct[b];
            if (own || (((dmask[sidx] >> b) & 1) && (directional(dx[sidx], dy[sidx], s, c) > thr))) {
                const uint32_t tagged = k | kTailTag;
                if (lane == 0) {
                    atomicExch(&label[sidx], tagged);
                    __hip_atomic_store(&queue[0], sidx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                int head = 0, tail = 1;
                while (head < tail) {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                    const int nsrc = min(8, tail - head);
                    bool claim = false;
                    int q = 0;
                    if (si < nsrc) {
                        const int p = __hip_atomic_load(&queue[head + si], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        q = p + noff;
                        const uint32_t lq = ld_agent(&label[q]);
                        if (lq == k) {  // a pixel it committed in the rounds, not yet visited
                            claim = atomicCAS(&label[q], k, tagged) == k;
                        } else if (lq == kLabelFree && ((dmask[q] >> b) & 1) && directional(dx[q], dy[q], s, c) > thr) {
                            claim = atomicCAS(&label[q], kLabelFree, tagged) == kLabelFree;
                        }
                    }
                    const uint64_t m = __ballot(claim);
                    if (claim)
                        __hip_atomic_store(&queue[tail + __popcll(m & ((1ull << lane) - 1ull))], q, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    tail += (int)__popcll(m);
                    head += nsrc;
                }
                size = tail;
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
                for (int i = lane; i < tail; i += 64) {
                    const int p = __hip_atomic_load(&queue[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(&label[p], k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
            }
        }
        if (lane == 0) seed_size[k] = size;
    }
}

}  // namespace

// ---- host side of the rounds ------------------------------------------------------------------

// LIBRECTIFY_FLOOD_DEBUG: what the round's exploration did (synchronises; rounds are then enqueued one at a time)
static void flood_debug_round(const FloodBuffers& B, const FloodFrame& F, uint32_t n_seeds, hipStream_t s) {
    uint32_t ctrl[kFloodCtrlWords];
    (void)hipStreamSynchronize(s);
    (void)hipMemcpy(ctrl, B.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost);
    const uint32_t n_act = ctrl[kCtrlNAct];
    if (n_act == 0 || ctrl[kCtrlGiantStep] != 0u) return;  // (nothing left, or a giant step is asked for: the round did nothing)
    const uint32_t* act = (ctrl[kCtrlRounds] & 1u) ? B.act_b : B.act_a;
    std::vector<uint32_t> cnt(n_seeds), blk(n_seeds), flg(n_seeds), actv(n_act);
    (void)hipMemcpy(flg.data(), B.flags, n_seeds * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipMemcpy(actv.data(), act, n_act * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipMemcpy(cnt.data(), B.count, n_seeds * sizeof(uint32_t), hipMemcpyDeviceToHost);
    (void)hipMemcpy(blk.data(), B.blocked, n_seeds * sizeof(uint32_t), hipMemcpyDeviceToHost);
#ifdef LR_REWALK_TIMING
    {
        unsigned long long t[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_rewalk_timing), sizeof(t));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_rewalk_timing), z, sizeof(z));
        if (t[0] > 0)
            std::fprintf(stderr, "  re-walks from logs: %llu seeds, %llu records; s_memtime ticks per seed: records in %.0f, pixels %.0f, components %.0f, "
                         "unions %.0f, footprint %.0f, stamps %.0f\n", t[0], t[1], (double)t[2] / t[0], (double)t[3] / t[0], (double)t[4] / t[0],
                         (double)t[5] / t[0], (double)t[6] / t[0], (double)t[7] / t[0]);
    }
#endif
#ifdef LR_WALK_TIMING
    {
        unsigned long long t[8], z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_walk_timing), sizeof(t));
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_walk_timing), z, sizeof(z));
        if (t[1] > 0)
            std::fprintf(stderr, "  walks of more than 100 steps: %llu, %llu steps (%llu of them on a known tile); s_memtime ticks per step: "
                         "wait for pixels %.1f, closure %.1f, table %.1f, push %.1f, pop + issue %.1f\n",
                         t[0], t[1], t[7], (double)t[2] / t[1], (double)t[3] / t[1], (double)t[4] / t[1], (double)t[5] / t[1],
                         (double)t[6] / t[1]);
    }
#endif
    unsigned long long tsteps = 0, tpx = 0;
    uint32_t mxs = 0, mxk = 0, nb = 0;
    for (uint32_t i = 0; i < n_act; ++i) {
        const uint32_t kk = actv[i], st_ = flg[kk] >> 8;
        tsteps += st_;
        tpx += cnt[kk];
        nb += blk[kk] != 0;
        if (st_ > mxs) {
            mxs = st_;
            mxk = kk;
        }
    }
    {  // how much of the walking belongs to seeds that commit in this round
        unsigned long long steps_commit = 0, steps_other = 0;
        uint32_t longest_commit = 0, longest_other = 0, n_commit = 0;
        for (uint32_t i = 0; i < n_act; ++i) {
            const uint32_t kk = actv[i], st_ = flg[kk] >> 8;
            if (cnt[kk] > 0 && blk[kk] == 0 && !(flg[kk] & (kFlagIncomplete | kFlagSelfFail)) && kk < ctrl[kCtrlBarrier]) {
                steps_commit += st_;
                n_commit += 1;
                longest_commit = std::max(longest_commit, st_);
            } else {
                steps_other += st_;
                longest_other = std::max(longest_other, st_);
            }
        }
        std::fprintf(stderr, "  %u committing seeds: %llu steps, longest %u; blocked or dying seeds: %llu steps, longest %u\n",
                     n_commit, steps_commit, longest_commit, steps_other, longest_other);
    }
    {  // seeds that end with this round because a committing seed takes their pixel: what their walks cost
        const size_t npix = (size_t)F.w * F.h;
        std::vector<uint32_t> lab(npix);
        std::vector<int32_t> sidx(n_seeds);
        (void)hipMemcpy(lab.data(), F.label, npix * sizeof(uint32_t), hipMemcpyDeviceToHost);
        (void)hipMemcpy(sidx.data(), F.seed_idx, n_seeds * sizeof(int32_t), hipMemcpyDeviceToHost);
        auto commits = [&](uint32_t kk) {
            return cnt[kk] > 0 && blk[kk] == 0 && !(flg[kk] & (kFlagIncomplete | kFlagSelfFail)) && kk < ctrl[kCtrlBarrier];
        };
        unsigned long long steps_dying = 0, steps_dying_own = 0;
        uint32_t n_dying = 0, n_own = 0, longest = 0;
        for (uint32_t i = 0; i < n_act; ++i) {
            const uint32_t kk = actv[i], st_ = flg[kk] >> 8;
            if (commits(kk)) continue;
            const uint32_t v = lab[(size_t)sidx[kk]];
            if (v >= kMarkBit && v != kLabelFree) {
                const uint32_t j = v & ~kMarkBit;
                if (j != kk && commits(j)) {
                    steps_dying += st_;
                    n_dying += 1;
                    longest = std::max(longest, st_);
                }
                if (j != kk) {  // a lower seed reaches the seed pixel, whether or not it commits
                    steps_dying_own += st_;
                    n_own += 1;
                }
            }
        }
        std::fprintf(stderr, "  %u seeds die with this round (a committing seed takes their pixel): %llu steps, longest %u; "
                     "%u seeds have their own pixel stamped by a lower seed: %llu steps\n",
                     n_dying, steps_dying, longest, n_own, steps_dying_own);
    }
    {  // walk-length histogram (steps) and where in the seed order the long walks sit
        const uint32_t edges[8] = {8, 16, 32, 48, 64, 128, 192, 0xFFFFFFFFu};
        uint32_t hist[8] = {0, 0, 0, 0, 0, 0, 0, 0}, long_lo = 0xFFFFFFFFu, nlong_strong = 0;
        for (uint32_t i = 0; i < n_act; ++i) {
            const uint32_t kk = actv[i], st_ = flg[kk] >> 8;
            for (int b = 0; b < 8; ++b)
                if (st_ <= edges[b]) {
                    hist[b]++;
                    break;
                }
            if (st_ > 64) {
                long_lo = std::min(long_lo, kk);
                if (kk < n_seeds / 10 * 8) nlong_strong++;
            }
        }
        std::fprintf(stderr, "  steps<=8:%u <=16:%u <=32:%u <=48:%u <=64:%u <=128:%u <=192:%u more:%u; lowest seed with >64 steps: %u; such seeds among the strongest 80%%: %u\n",
                     hist[0], hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], long_lo, nlong_strong);
    }
    std::fprintf(stderr,
                 "flood round %u: active %u, %llu px walked in %llu steps, longest walk %u steps (%u px, seed %u), "
                 "blocked %u, barrier %u, slabs %u\n",
                 ctrl[kCtrlRounds] + 1, n_act, tpx, tsteps, mxs, cnt[mxk], mxk, nb, ctrl[kCtrlBarrier],
                 ctrl[kCtrlSlabs]);
    {  // the longest walk's seed: storage tier marks and log
        uint8_t t = 0;
        uint32_t ll = 0, full[kCtrlWords];
        (void)hipMemcpy(&t, B.tier + mxk, 1, hipMemcpyDeviceToHost);
        if (B.log_len && mxk < B.log_seeds) (void)hipMemcpy(&ll, B.log_len + mxk, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(full, B.ctrl, sizeof(full), hipMemcpyDeviceToHost);
        std::fprintf(stderr, "  ... that seed: tier marks %u (1 outgrew the first tier, 4 held as a giant), log %u records%s; window %u, lowest active %u, lowest held %u, log records handed out %u of %u\n",
                     (unsigned)t, ll & 0x7FFFFFFFu, (ll >> 31) ? " (cut down)" : "", full[kCtrlWindow], full[kCtrlLowest], full[kCtrlGiantLow], full[kCtrlLogTotal], B.log_cap);
    }
}

namespace {

const bool g_flood_debug = std::getenv("LIBRECTIFY_FLOOD_DEBUG") != nullptr;

FloodArgs flood_args(const FloodBuffers& B, const FloodFrame& F, bool use_big) {
    FloodArgs A;
    A.dx = F.dx;
    A.dy = F.dy;
    A.dmask = F.dmask;
    A.w = F.w;
    A.h = F.h;
    A.tiles_x = (F.w + 7) / 8;
    A.seed_idx = F.seed_idx;
    A.seed_bin = F.seed_bin;
    A.seed_thr = F.seed_thr;
    A.label = F.label;
    A.blocked = B.blocked;
    A.count = B.count;
    A.flags = B.flags;
    A.tier = B.tier;
    A.ctrl = B.ctrl;
    A.dirty = B.dirty;
    A.slab_ring = (uint4*)B.slab_ring;
    A.slab_hash = (uint4*)B.slab_hash;
    A.n_slabs = B.n_slabs;
    A.slab_ring_cap = B.slab_ring_cap;
    A.slab_hash_cap = B.slab_hash_cap;
    static const int order_env = std::getenv("LIBRECTIFY_FLOOD_ORDER") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_ORDER")) : -1;
    A.from_end = order_env >= 0 ? (uint32_t)order_env : 1u;
    A.no_rest = 0u;
    A.next_reach = 0xFFFFFFFFu;
    A.win_shift = 2u;
    const uint32_t big_cap = B.big_cap_override ? B.big_cap_override : kBigCap;
    A.big_cap = use_big ? big_cap : 0u;
    static const int t1_env = std::getenv("LIBRECTIFY_FLOOD_T1_TILES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_T1_TILES")) : 0;
    // With the giants' rule a second-tier walk counts as a giant at 1 024 tiles, two thirds of what the team's table holds
    // (LIBRECTIFY_FLOOD_TEAM_TILES; 0 = the table's 1 536): ramp frame 11.0 -> 8.7 ms, regions 41.5 -> 39.6, frames without
    // such walks unchanged; at 640 the edge-less 4K frame pays (5.7 -> 7.0 ms).
    static const int team_tiles_env = std::getenv("LIBRECTIFY_FLOOD_TEAM_TILES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_TEAM_TILES")) : 1024;
    A.team_tiles = B.team_tile_cap ? B.team_tile_cap : ((B.giant_hold && team_tiles_env > 0) ? (uint32_t)team_tiles_env : 0xFFFFFFFFu);
    static const int many_env = std::getenv("LIBRECTIFY_FLOOD_GIANT_MANY") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_GIANT_MANY")) : 64;
    A.giant_many = B.giant_hold ? (uint32_t)std::max(many_env, 0) : 0u;
    static const int defer_env = std::getenv("LIBRECTIFY_FLOOD_DEFER_STEPS") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_DEFER_STEPS")) : 512;
    A.blk = (B.giant_hold && defer_env > 0) ? B.blk : nullptr;
    A.defer_steps = (uint32_t)std::max(defer_env, 1);
    A.handover = B.handover;
    A.t1_tiles = t1_env > 8 ? (uint32_t)t1_env : 0xFFFFFFFFu;
    static const int t1r_env = std::getenv("LIBRECTIFY_FLOOD_T1_REGIONAL") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_T1_REGIONAL")) : 32;
    static const int t1m_env = std::getenv("LIBRECTIFY_FLOOD_T1_REGIONAL_MIN") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_T1_REGIONAL_MIN")) : 16;
    A.t1_regional = t1r_env > 8 ? (uint32_t)t1r_env : 0xFFFFFFFFu;
    A.t1_regional_min = (uint32_t)std::max(t1m_env, 1);
    static const int holdmin_env = std::getenv("LIBRECTIFY_FLOOD_HOLD_MIN") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_HOLD_MIN")) : 4;
    A.hold_min_big = (uint32_t)std::max(holdmin_env, 1);
    static const int holdrel_env = std::getenv("LIBRECTIFY_FLOOD_HOLD_RELEASE") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_HOLD_RELEASE")) : 64;
    A.hold_release = (uint32_t)std::max(holdrel_env, 0);
    static const int t1w_env = std::getenv("LIBRECTIFY_FLOOD_T1_WIDE_TILES") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_T1_WIDE_TILES")) : 0;
    static const int t1f_env = std::getenv("LIBRECTIFY_FLOOD_T1_WIDE_FRONT") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_T1_WIDE_FRONT")) : 6;
    A.t1_wide_tiles = t1w_env > 0 ? (uint32_t)t1w_env : 0xFFFFFFFFu;
    A.t1_wide_front = (uint32_t)std::max(t1f_env, 1);
    // way-points: walks of this many tiles leave them (LIBRECTIFY_FLOOD_MULTI_MIN; profiles/r04_flood_multi_sweep.txt)
    static const int wp_min_env = std::getenv("LIBRECTIFY_FLOOD_MULTI_MIN") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_MULTI_MIN")) : 100;
    A.waypoints = B.waypoints;
    A.wp_cap = B.waypoints ? B.wp_cap : 0u;
    A.multi_list = B.multi_list;
    A.multi_next = 0u;
    A.wp_min_tiles = (B.multi_source && B.waypoints && use_big) ? (uint32_t)std::max(wp_min_env, (int)kWpK + 1) : 0xFFFFFFFFu;
    // logs: walks of this many tiles leave one (LIBRECTIFY_FLOOD_LOG_MIN); with logs there are no way-points
    static const int log_min_env = std::getenv("LIBRECTIFY_FLOOD_LOG_MIN") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOG_MIN")) : 16;
    const bool logs = B.rewalk_logs && B.log_buf && B.log_off && B.log_len && B.multi_list;
    A.log_min_tiles = logs ? (uint32_t)std::max(B.log_min_tiles > 0 ? B.log_min_tiles : log_min_env, 1) : 0xFFFFFFFFu;
    static const int log_walk_env = std::getenv("LIBRECTIFY_FLOOD_LOG_WALK") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOG_WALK")) : 12;
    A.log_walk_tiles = (uint32_t)std::max(B.log_walk_tiles > 0 ? B.log_walk_tiles : log_walk_env, 3);
    static const bool log_sweep_env = std::getenv("LIBRECTIFY_FLOOD_LOG_SWEEP") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_LOG_SWEEP")) != 0;
    A.log_sweep = (log_sweep_env || B.log_sweep) ? 1u : 0u;
    A.host_progress = (B.jit_first > 0 && !g_flood_debug) ? B.host_progress : nullptr;
    static const bool mirror_off = std::getenv("LIBRECTIFY_FLOOD_MIRROR") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_MIRROR")) == 0;  // (comparison)
    A.host_ctrl = (A.host_progress && !mirror_off) ? B.host_ctrl : nullptr;
    A.quiet = 0u;
    A.giant_hold = B.giant_hold ? 1u : 0u;
    A.log_max_len = (B.rewalk_big && use_big) ? (uint32_t)kRewalkTilesBig : (uint32_t)kRewalkTiles;
    A.log_seeds = logs ? B.log_seeds : 0u;
    A.log_off = B.log_off;
    A.log_len = B.log_len;
    A.log_buf = B.log_buf;
    A.log_cap = B.log_cap;
    if (logs) A.wp_min_tiles = 0xFFFFFFFFu;
    static const int g_cap_env = std::getenv("LIBRECTIFY_FLOOD_PARTIAL_STEPS") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_PARTIAL_STEPS")) : 16;
    A.g_cap = g_cap_env > 0 ? (uint32_t)g_cap_env : kMaxSteps;
    A.act_a = B.act_a;
    A.act_b = B.act_b;
    A.giant_mask = reinterpret_cast<unsigned long long*>(B.giant_mask);
    A.giant_parent = B.giant_parent;
    A.giant_step = (B.giant_hold && B.giant_step && B.giant_mask && B.giant_parent && use_big) ? 1u : 0u;
    return A;
}


// one round: explore (main launch, the entries past its grid, second LDS tier), commit pass, survivors pass
// `known_len`: the length of this round's list when the host has seen it (rounds enqueued just in time), 0xFFFFFFFF when the
// round is enqueued blindly; `next_known`: the NEXT round will be enqueued with its length known.  A round whose list is
// known to be longer than its grid gets the `rest` launch whatever its index, and a round whose successor will know needs
// no barrier for entries the successor "will not reach".  (Lists stay long into the late rounds when a window is closed in
// front of waiting seeds -- the hold-back, the giants' line: a late round that walked only its first grid's worth of an
// unordered list could leave the lowest active seed unwalked, move nothing, and send the frame to the ordered tail.)
constexpr uint32_t kFullGridCap = 1u << 18;
constexpr int kRestRounds = 4;  // rounds 0 .. kRestRounds - 1 always bring their `rest` launch: the rounds a frame may enqueue blindly (context.hip: jit_first_max)
void enqueue_round(const FloodBuffers& B, const FloodFrame& F, const FloodArgs& A0, bool use_big, int index, hipStream_t s,
                   uint32_t known_len = 0xFFFFFFFFu, bool next_known = false) {
    // (which of the two active lists a round reads is the device's business -- kCtrlRounds, act_now: rounds enqueued behind a
    // request for a giant step do nothing and do not count; `index` only sizes grids and picks the launches of a round)
    const size_t npix = (size_t)F.w * F.h;
    const int pix_blocks = (int)std::min<size_t>((npix + 255) / 256, 4096);
    const int seed_blocks = (int)std::min<uint32_t>((F.seed_cap + 255) / 256, 256);
    hipEvent_t dbg0 = nullptr, dbg1 = nullptr;
    if (g_flood_debug) {
        (void)hipEventCreate(&dbg0);
        (void)hipEventCreate(&dbg1);
        (void)hipEventRecord(dbg0, s);
    }
    // grid: see flood_explore_kernel.  A staged start keeps the list long for a round more.
    // (not below a quarter: while the weakest fifth of the seeds is held back, the list stays that long)
    // From the fifth round on (kRestRounds) the lists are a tenth of the grid (a quarter of the capacity) and the `rest` launch -- the
    // entries past the guess -- was one empty launch per round, blind rounds included.  It is gone there; should such a list
    // ever be longer than its grid, the round walks its first entries and the survivors pass that wrote the list has set
    // the round's barrier at the lowest seed behind them (next_reach): nothing above an unwalked seed commits, exact as
    // with a walk that ran out of storage.  (Test hook LIBRECTIFY_FLOOD_TEST_GRID: a tiny grid for those rounds.)
    static const int test_grid = std::getenv("LIBRECTIFY_FLOOD_TEST_GRID") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_TEST_GRID")) : 0;
    auto grid_of = [&](int idx) {
        const int shift = std::min(std::max(idx - (B.win_first_shift > 0 ? 1 : 0), 0), 2);
        uint32_t g = std::max<uint32_t>(std::min<uint32_t>(F.seed_cap, 2048u), F.seed_cap >> shift);
        // Round 5: up to kFullGridCap seeds the guess is the whole capacity in every blind round -- 30 000 workgroups that leave
        // at once cost round two 2.4 us, the `rest` launch they make unnecessary cost three rounds 4.7 us each (and a blind
        // round's launch waits for room beside the other lanes' walks); the rounds enqueued just in time know their length.
        if (F.seed_cap <= kFullGridCap) g = F.seed_cap;
        if (test_grid > 0 && idx >= kRestRounds) g = std::min<uint32_t>(g, (uint32_t)test_grid);
        return g;
    };
    auto has_rest = [&](int idx) { return idx < kRestRounds && grid_of(idx) < F.seed_cap; };
    // (a round enqueued with its list's length known takes a grid of exactly that many workgroups: the guess is cap / 4 in the
    // late rounds, some 14 000 workgroups that leave at once for a list of a few hundred, and the dispatcher's time is the
    // one thing all the frames in flight share)
    const bool exact = known_len < 0xFFFFFFFEu && known_len <= F.seed_cap && !(test_grid > 0 && index >= kRestRounds);
    const uint32_t grid = exact ? std::max<uint32_t>(known_len, 1u) : grid_of(index);
    const bool rest_now = !exact && (has_rest(index) || (known_len != 0xFFFFFFFFu && known_len > grid));
    FloodArgs A = A0;
    A.no_rest = (grid < F.seed_cap && !rest_now && !exact) ? 1u : 0u;
    A.next_reach = (grid_of(index + 1) < F.seed_cap && !has_rest(index + 1) && !next_known) ? grid_of(index + 1) : 0xFFFFFFFFu;
    // Way-point seeds listed by the last survivors pass are walked by teams in a launch of their own on the context's second
    // stream, BESIDE this round's exploration (one stream runs its kernels one after the other, and hipExtAnyOrderLaunch is
    // not honoured on gfx950: tools/ubench/anyorder.hip): fork behind the last round, join in front of the commit passes.
    // Only in the rounds that have such walks to speak of (2 .. kMultiRoundLast + 1): a fork and a join are two event
    // packets a round.
    const bool multi_now = B.aux_stream != nullptr && A.wp_min_tiles != 0xFFFFFFFFu && use_big && index >= 1 && index <= B.multi_round_last &&
                           index < B.n_fork_events && !g_flood_debug;
    A.multi_next = (B.aux_stream != nullptr && A.wp_min_tiles != 0xFFFFFFFFu && use_big && index + 1 >= 1 && index + 1 <= B.multi_round_last &&
                    index + 1 < B.n_fork_events && !g_flood_debug) ? 1u : 0u;
    const bool logs = A.log_min_tiles != 0xFFFFFFFFu;
    A.log_use = (logs && index >= B.log_from_round) ? 1u : 0u;
    if (multi_now) {
        (void)hipEventRecord(B.fork_events[index], s);
        (void)hipStreamWaitEvent(B.aux_stream, B.fork_events[index], 0);
        hipLaunchKernelGGL(flood_explore_team_kernel, dim3(std::min<uint32_t>(F.seed_cap, kTeamGrid)), dim3(64 * kTeamWaves),
                           kTeamLdsBytes, B.aux_stream, A, F.trig, B.multi_list, 1u);
        (void)hipEventRecord(B.join_events[index], B.aux_stream);
    }
    hipLaunchKernelGGL(flood_explore_kernel, dim3(grid), dim3(64), 0, s, A, F.trig, B.big_list);
    if (rest_now)  // entries past the guess, if any
        hipLaunchKernelGGL(flood_explore_rest_kernel, dim3(1024), dim3(64), 0, s, A, F.trig, B.big_list, grid);
    static const bool team = !(std::getenv("LIBRECTIFY_FLOOD_TEAM") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_TEAM")) == 0);
    if (use_big && team)
        hipLaunchKernelGGL(flood_explore_team_kernel, dim3(std::min<uint32_t>(F.seed_cap, kTeamGrid)), dim3(64 * kTeamWaves),
                           kTeamLdsBytes, s, A, F.trig, B.big_list, 0u);
    else if (use_big)
        hipLaunchKernelGGL(flood_explore_big_kernel, dim3(std::min<uint32_t>(F.seed_cap, kBigCap)), dim3(64), kBigLdsBytes, s,
                           A, F.trig, B.big_list);
    if (multi_now) (void)hipStreamWaitEvent(s, B.join_events[index], 0);
    if (logs && index >= std::max(B.log_from_round, 1))
        hipLaunchKernelGGL((flood_rewalk_kernel<kRewalkThreads, kRewalkTiles>), dim3(std::min<uint32_t>(F.seed_cap, kRewalkGrid)), dim3(kRewalkThreads),
                           rewalk_lds_bytes<kRewalkTiles>(), s, A, B.multi_list, 1u);
    if (logs && index >= std::max(B.log_from_round, 1) && A.log_max_len > (uint32_t)kRewalkTiles)
        hipLaunchKernelGGL((flood_rewalk_kernel<kRewalkThreadsBig, kRewalkTilesBig>), dim3(std::min<uint32_t>(F.seed_cap, 1024u)),
                           dim3(kRewalkThreadsBig), rewalk_lds_bytes<kRewalkTilesBig>(), s, A, B.multi_list, (uint32_t)kRewalkTiles + 1u);
    if (g_flood_debug) (void)hipEventRecord(dbg1, s);
    if (g_flood_debug) {
        uint32_t n = 0;
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(&n, F.d_n_seeds, sizeof(n), hipMemcpyDeviceToHost);
        flood_debug_round(B, F, std::min(n, F.seed_cap), s);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, dbg0, dbg1);
        std::fprintf(stderr, "  explore kernels of this round: %.1f us\n", ms * 1e3f);
        (void)hipEventDestroy(dbg0);
        (void)hipEventDestroy(dbg1);
    }
    static const int g_rounds_env = std::getenv("LIBRECTIFY_FLOOD_PARTIAL_ROUNDS") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_PARTIAL_ROUNDS")) : 3;
    // (only in the first three rounds: with the logs the later rounds' re-walks are cheap, and what the partial commits save
    // there no longer pays their launch -- 0.970 -> 0.935 ms over the four bench frames, same rounds; and the lanes of a batch
    // are better off with a launch less per late round: 9.81 -> 9.99 Gpix/s)
    if (B.partial_commits && index < g_rounds_env)
        hipLaunchKernelGGL(flood_partial_commit_kernel, dim3(grid), dim3(64), 0, s, A, const_cast<uint8_t*>(F.dmask));
    hipLaunchKernelGGL(flood_commit_pixels_kernel, dim3(pix_blocks), dim3(256), 0, s, A, F.label, npix,
                       const_cast<uint8_t*>(F.dmask));
    hipLaunchKernelGGL(flood_survivors_kernel, dim3(seed_blocks), dim3(256), 0, s, A, B.state, F.seed_size);
}

// The giant step (kCtrlGiantStep): four launches sized by the frame's tiles; each leaves at once when no step is asked for.
void enqueue_giant_step(const FloodBuffers& B, const FloodFrame& F, const FloodArgs& A, hipStream_t s) {
    const uint32_t tiles_x = (uint32_t)(F.w + 7) / 8, tiles_y = (uint32_t)(F.h + 7) / 8;
    const uint32_t strips = ((tiles_x + 7) / 8) * tiles_y, runs = ((tiles_x + 63) / 64) * tiles_y, n_tiles = tiles_x * tiles_y;
    hipEvent_t dbg0 = nullptr, dbg1 = nullptr;
    if (g_flood_debug) {
        (void)hipEventCreate(&dbg0);
        (void)hipEventCreate(&dbg1);
        (void)hipEventRecord(dbg0, s);
    }
    hipLaunchKernelGGL(giant_mask_kernel, dim3(std::min<uint32_t>((strips + 3) / 4, 8192u)), dim3(256), 0, s, A, F.trig);
    hipLaunchKernelGGL(giant_merge_kernel, dim3(std::min<uint32_t>((n_tiles + 255) / 256, 4096u)), dim3(256), 0, s, A);
    hipLaunchKernelGGL(giant_commit_kernel, dim3(std::min<uint32_t>((runs + 3) / 4, 4096u)), dim3(256), 0, s, A, const_cast<uint8_t*>(F.dmask));
    hipLaunchKernelGGL(giant_finish_kernel, dim3(std::min<uint32_t>((F.seed_cap + 255) / 256, 256u)), dim3(256), 0, s, A);
    if (g_flood_debug) {
        uint32_t ctrl[kCtrlWords];
        (void)hipEventRecord(dbg1, s);
        (void)hipStreamSynchronize(s);
        (void)hipMemcpy(ctrl, B.ctrl, sizeof(ctrl), hipMemcpyDeviceToHost);
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, dbg0, dbg1);
        std::fprintf(stderr, "giant step %u: %.1f us; lowest active seed now %u, lowest held %u, window %u, next step for seed %d\n", ctrl[kCtrlGiantDone],
                     ms * 1e3f, ctrl[kCtrlLowest], ctrl[kCtrlGiantLow], ctrl[kCtrlWindow], (int)ctrl[kCtrlGiantStep] - 1);
        (void)hipEventDestroy(dbg0);
        (void)hipEventDestroy(dbg1);
    }
}

}  // namespace

// Rounds are enqueued blindly: their kernels read the list length from the control block, and a round enqueued past
// the end does nothing.  Typical frames finish within the first batch, so the flood needs no host synchronisation of
// its own: the caller goes on enqueuing the later stages and looks at the control block when the frame is done.
int flood_enqueue(const FloodBuffers& B, const FloodFrame& F, FloodProgress* P, uint32_t* h_ctrl, hipStream_t s) {
    P->enqueued = 0;
    P->sizes_known = false;
    P->max_flood = 0;
    P->use_big = B.second_tier && B.second_tier_from_start;
    if (F.seed_cap == 0) return launch_label_init(F.label, (size_t)F.w * F.h, s);
    // staged start (FloodBuffers::win_*); LIBRECTIFY_FLOOD_WINDOW="<first shift>,<growth shift>" overrides
    static const char* win_env = std::getenv("LIBRECTIFY_FLOOD_WINDOW");
    int win_first_shift = B.win_first_shift, win_growth = B.win_growth;
    if (B.staged_from_start && win_first_shift <= 0) {  // the strongest quarter first, twice as many a round (kCtrlStaged)
        win_first_shift = 2;
        win_growth = 1;
    }
    if (win_env) {
        win_first_shift = std::atoi(win_env);
        const char* c = std::strchr(win_env, ',');
        if (c) win_growth = std::max(1, std::atoi(c + 1));
    }
    P->win_growth = win_growth;
    static const int hold_env = std::getenv("LIBRECTIFY_FLOOD_HOLD") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_HOLD")) : -1;
    const int hold_pct = hold_env >= 0 ? hold_env : B.win_hold_pct;
    static const bool hold_start_env = std::getenv("LIBRECTIFY_FLOOD_HOLD_START") != nullptr;
    const bool hold_start = B.hold_from_start || hold_start_env;
    // the opt-in for more than 32 KB of dynamic LDS is a per-device attribute of the kernel: once per device of this process
    {
        static std::atomic<uint64_t> done_mask{0};
        int dev = 0;
        LR_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_acquire) & bit)) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(flood_explore_big_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kBigLdsBytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(flood_explore_team_kernel),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)kTeamLdsBytes) != hipSuccess ||
                hipFuncSetAttribute(reinterpret_cast<const void*>(flood_rewalk_kernel<kRewalkThreadsBig, kRewalkTilesBig>),
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)rewalk_lds_bytes<kRewalkTilesBig>()) != hipSuccess) {
                set_error("flood: cannot reserve the second-tier LDS");
                return 1;
            }
            done_mask.fetch_or(bit, std::memory_order_release);
        }
    }
    {
        static const int dense_env = std::getenv("LIBRECTIFY_FLOOD_DENSE_DIV") ? std::atoi(std::getenv("LIBRECTIFY_FLOOD_DENSE_DIV")) : 12;
        const uint32_t dense_div = (uint32_t)std::max(dense_env, 0);
        const size_t npix = (size_t)F.w * F.h;
        const uint32_t seed_blocks = (F.seed_cap + 255) / 256;
        const uint32_t blocks = std::max<uint32_t>(seed_blocks, (uint32_t)std::min<size_t>((npix + 255) / 256, 4096));
        hipLaunchKernelGGL(flood_init_seeds_kernel, dim3(blocks), dim3(256), 0, s, F.d_n_seeds, F.seed_cap, B.act_a, B.state,
                           B.tier, B.blocked, B.count, B.flags, F.seed_size, B.ctrl, B.dirty,
                           (uint32_t)((npix + 255) >> 8), win_first_shift, hold_pct, hold_start ? 1u : 0u, F.label, npix,
                           B.waypoints, B.wp_cap, B.log_len, B.log_len ? B.log_seeds : 0u, dense_div,
                           (B.staged_from_start && !win_env && B.win_first_shift <= 0) ? 1u : 0u, B.blk);
    }
    FloodArgs A = flood_args(B, F, P->use_big);
    A.win_shift = (uint32_t)win_growth;
    bool ended_in_sight = false;
    // rounds enqueued blindly: two more than the context's previous frame needed (a round past the end costs five
    // empty launches, a round too few costs the frame a second lap through the fit and the grouping)
    if (A.host_progress) {
        // just in time (FloodBuffers::host_progress): the rounds the last frame needed less one at once, every further one when
        // the host has seen the rounds so far leave seeds.  A round that never reports (nothing should keep it) ends the
        // watch after a second: the rest goes the blind way, flood_finish picks up whatever is left.
        // (one 64-bit word, so that every look is a consistent report: flood_report)
        unsigned long long* hp = reinterpret_cast<unsigned long long*>(B.host_progress);
        __atomic_store_n(hp, 0ull, __ATOMIC_SEQ_CST);
        struct Report {
            bool any;
            uint32_t n_left, rounds, giants;
            bool stalled, want_giant, huge, calm;
        };
        auto look = [&]() {
            const unsigned long long w = __atomic_load_n(hp, __ATOMIC_ACQUIRE);
            return Report{(w >> 63) != 0ull, (uint32_t)(w & 0x1FFFFFFFull), (uint32_t)((w >> 32) & 0xFFFull), (uint32_t)((w >> 44) & 0xFFFFull),
                          ((w >> 29) & 1ull) != 0ull, ((w >> 30) & 1ull) != 0ull, ((w >> 31) & 1ull) != 0ull, ((w >> 60) & 1ull) != 0ull};
        };
        const int first = std::min(std::max(B.jit_first, 1), 16);
        // (LIBRECTIFY_FLOOD_JIT_LEAD=1 keeps one round ahead -- the next round goes in when all but the last one enqueued are
        // over and left seeds, the host's reaction hides behind that last round, at most one round is enqueued in vain:
        // measured the same as none ahead, 0.921 against 0.923 ms over eight 4K frames, blind rounds 0.937)
        const int lead = B.jit_lead;
        // The context's last frame never needed the second tier (calm_hint): the rounds enqueued blindly behind the first go
        // without its launch -- empty on such frames, and in a batch each waits ~50 us for room beside the other lanes' walks.
        // A walk that outgrows the first tier there counts as unfinished (explore_seed: A.quiet), the report stops saying
        // "calm", and the rounds enqueued from then on bring the second tier.
        const bool multi_on0 = B.aux_stream != nullptr && A.wp_min_tiles != 0xFFFFFFFFu;
        const bool quiet = B.calm_hint && P->use_big && !multi_on0;
        FloodArgs A_quiet = A;
        if (quiet) {
            A_quiet = flood_args(B, F, false);
            A_quiet.win_shift = A.win_shift;
            A_quiet.quiet = 1u;
        }
        for (int r = 0; r < first; ++r, ++P->enqueued) {
            const bool q = quiet && r >= 1;
            enqueue_round(B, F, q ? A_quiet : A, q ? false : P->use_big, P->enqueued, s, 0xFFFFFFFFu, r == first - 1);
        }
        // Rounds that count on the device (kCtrlRounds) against rounds enqueued that can still count: a round enqueued behind a
        // request for a giant step does nothing, so once a request is seen every round enqueued so far is accounted for.
        int counting = P->enqueued;
        uint32_t giants = 0;  // giant steps enqueued
        bool calm = false;
        FloodArgs A_calm = A;
        const auto t0 = std::chrono::steady_clock::now();
        int spins = 0;
        // (a single call spins -- its flood is a millisecond -- but not for ever: content that takes the flood tens of
        // milliseconds should not cost the caller a core.  After two milliseconds it yields between looks, after twenty it
        // sleeps 20 us between them like the lanes of a batch.  A sleep of 20 us lasts 70 with the default timer slack of
        // 50 us -- measured as 45-77 us of idle stream between the rounds of a slow frame, eighteen times on a frame of soft
        // blobs -- so the thread's slack is a microsecond while it watches and what it was afterwards.)
        bool slow = false, very_slow = false;
        const int slack_before = (B.jit_sleep_us > 0) ? prctl(PR_GET_TIMERSLACK) : -1;
        int slack_set = -1;
        auto short_slack = [&]() {
            if (slack_set >= 0) return;
            slack_set = slack_before >= 0 ? slack_before : prctl(PR_GET_TIMERSLACK);
            if (slack_set > 1000) (void)prctl(PR_SET_TIMERSLACK, 1000UL);
            else slack_set = -2;  // (short already, or not to be had: nothing to restore)
        };
        if (B.jit_sleep_us > 0) short_slack();
        struct SlackRestore {
            int* v;
            ~SlackRestore() {
                if (*v > 1000) (void)prctl(PR_SET_TIMERSLACK, (unsigned long)*v);
            }
        } slack_restore{&slack_set};
        auto deadline_passed = [&]() {
            if (B.jit_sleep_us > 0 || very_slow) std::this_thread::sleep_for(std::chrono::microseconds(B.jit_sleep_us > 0 ? B.jit_sleep_us : 20));
            else if (slow) std::this_thread::yield();
            if ((++spins & ((B.jit_sleep_us > 0 || slow) ? 15 : 1023)) != 0) return false;
            const auto dt = std::chrono::steady_clock::now() - t0;
            if (dt > std::chrono::milliseconds(2)) slow = true;
            if (dt > std::chrono::milliseconds(20) && !very_slow) {
                very_slow = true;
                short_slack();
            }
            return dt > std::chrono::seconds(1);
        };
        for (;;) {
            Report r{};
            bool timed_out = false;
            for (;;) {
                r = look();
                if (r.any && ((int)r.rounds + lead >= counting || r.n_left == 0u || r.want_giant)) break;
                if (deadline_passed()) {
                    timed_out = true;
                    break;
                }
            }
            // The lowest active seed is a marked giant: the whole device floods it (giant_*_kernel), and the next one if the
            // step's last kernel asks for it, before the next round with work goes in.
            // (A step's last kernel asks for the next one when the next lowest seed is a marked giant too -- a ring after another,
            // a stripe after another.  In such a chain the steps go in two, four, eight at a time: a step of a chain that reuses
            // the union-find is 24 us of kernels, and the hand-over to the host and back was 21 us between every two of them.
            // Steps enqueued behind the chain's end leave at once.)
            uint32_t burst = 1u;
            while (!timed_out && r.want_giant && r.n_left != 0u && !r.stalled) {
                const uint32_t base = r.giants;
                for (uint32_t b2 = 0; b2 < burst; ++b2) enqueue_giant_step(B, F, A, s);
                giants += burst;
                for (;;) {
                    r = look();
                    if (r.giants >= base + burst || (r.giants > base && !r.want_giant)) break;
                    if (deadline_passed()) {
                        timed_out = true;
                        break;
                    }
                }
                counting = (int)r.rounds;
                if (giants >= 4096u) break;  // (never: a step retires a seed)
                burst = std::min<uint32_t>(burst * 2u, 8u);
            }
            if (timed_out) {
                // (with the `rest` launch whatever the list's length: the round before them was told its successor would know)
                for (int k = 0; k < 2; ++k, ++P->enqueued) enqueue_round(B, F, A, P->use_big, P->enqueued, s, 0xFFFFFFFEu, true);
                break;
            }
            if (r.n_left == 0u && !r.stalled) {  // the flood is over, and the report says whether it committed a huge flood
                P->sizes_known = true;
                P->max_flood = r.huge ? 0xFFFFFFFFu : 0u;
                ended_in_sight = true;
            }
            if (r.n_left == 0u || r.stalled || P->enqueued >= 256) break;
            // A calm frame (flood_report): the rounds from here on go without the second tier's launch -- an empty launch of
            // 512 workgroups of 512 threads and 41 KB of LDS each, which in a batch waits ~50 us for room beside the other
            // lanes' walks.  (A walk that outgrew the first tier after all would move into a slab or count as unfinished:
            // exact either way.)
            static const bool calm_off = std::getenv("LIBRECTIFY_FLOOD_CALM") && std::atoi(std::getenv("LIBRECTIFY_FLOOD_CALM")) == 0;
            // (not with multi-source re-walks on: their launch is the second tier's, and the round before has listed seeds for it)
            const bool multi_on = B.aux_stream != nullptr && A.wp_min_tiles != 0xFFFFFFFFu;
            if (r.calm && !calm_off && !calm && P->use_big && !multi_on) {
                calm = true;
                A_calm = flood_args(B, F, false);
                A_calm.win_shift = A.win_shift;
            }
            if (calm) enqueue_round(B, F, A_calm, false, P->enqueued, s, r.n_left, true);
            else enqueue_round(B, F, A, P->use_big, P->enqueued, s, r.n_left, true);
            ++P->enqueued;
            ++counting;
        }
    } else {
    const int batch = g_flood_debug ? 1 : std::min(std::max(B.blind_rounds, 1), 16);
    for (int r = 0; r < batch; ++r, ++P->enqueued) enqueue_round(B, F, A, P->use_big, P->enqueued, s);
    }
    // (the host saw the flood end: the round that ended it has written the control block to B.host_ctrl itself)
    if (!(ended_in_sight && A.host_ctrl == h_ctrl))
        LR_HIP(hipMemcpyAsync(h_ctrl, B.ctrl, kCtrlWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    LR_HIP(hipGetLastError());
    return 0;
}

// h_ctrl holds the control block as flood_enqueue's copy delivered it (the stream has been synchronised since).
// Runs whatever is left: more rounds (three at a time, one synchronisation each), and the ordered tail if the rounds
// stalled on exhausted storage.  *extra tells the caller that the label image changed after its later stages ran.
int flood_finish(const FloodBuffers& B, const FloodFrame& F, FloodProgress* P, uint32_t* h_ctrl, int* rounds_out,
                 uint32_t* tiers_out, bool* extra, hipStream_t s) {
    *extra = false;
    *rounds_out = 0;
    if (F.seed_cap == 0) return 0;
    const uint32_t big_cap = B.big_cap_override ? B.big_cap_override : kBigCap;
    FloodArgs A = flood_args(B, F, P->use_big);
    A.win_shift = (uint32_t)P->win_growth;
    while (h_ctrl[kCtrlNAct] != 0u) {
        *extra = true;
        P->sizes_known = false;
        static const bool call_debug = std::getenv("LIBRECTIFY_CALL_DEBUG") != nullptr;
        if (g_flood_debug || call_debug)
            std::fprintf(stderr, "flood after %d rounds enqueued: %u done, active %u, remain %u, stall %u\n", P->enqueued,
                         h_ctrl[kCtrlRounds], h_ctrl[kCtrlNAct], h_ctrl[kCtrlNRemain], h_ctrl[kCtrlStall]);
        if (!P->use_big && B.second_tier && h_ctrl[kCtrlSlabTotal] > 0u) {  // long walks after all: second tier from now on
            P->use_big = true;
            A.big_cap = big_cap;
        }
        const int batch = g_flood_debug ? 1 : 3;
        // (a giant step was asked for and nobody was looking: the rounds behind the request did nothing.  Steps that follow
        // each other -- a step's last kernel asks for the next -- take a lap each here: this is the path of the debug mode,
        // of blind rounds and of a watch that timed out)
        if (h_ctrl[kCtrlGiantStep] != 0u && A.giant_step != 0u) enqueue_giant_step(B, F, A, s);
        // (these rounds always bring their `rest` launch: the rounds before them may have been told that their successor would
        // know its list's length, and a flood that is still going after its first laps has long lists to walk)
        for (int r = 0; r < batch; ++r, ++P->enqueued) enqueue_round(B, F, A, P->use_big, P->enqueued, s, 0xFFFFFFFEu, true);
        LR_HIP(hipMemcpyAsync(h_ctrl, B.ctrl, kCtrlWords * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
        LR_HIP(hipStreamSynchronize(s));
    }
    int rounds = (int)h_ctrl[kCtrlRounds];
    if (h_ctrl[kCtrlStall] != 0u && h_ctrl[kCtrlNRemain] > 0u) {
        // storage exhausted on the lowest active seed: finish in order (always exact); the list is unordered
        *extra = true;
        const uint32_t n_rem = h_ctrl[kCtrlNRemain];
        uint32_t* lists[2] = {B.act_a, B.act_b};
        uint32_t* act = lists[rounds & 1];  // what the last round with work appended to
        {
            std::vector<uint32_t> tmp(n_rem);
            LR_HIP(hipMemcpy(tmp.data(), act, n_rem * sizeof(uint32_t), hipMemcpyDeviceToHost));
            std::sort(tmp.begin(), tmp.end());
            LR_HIP(hipMemcpy(act, tmp.data(), n_rem * sizeof(uint32_t), hipMemcpyHostToDevice));
        }
        hipLaunchKernelGGL(flood_ordered_tail_kernel, dim3(1), dim3(64), 0, s, F.dx, F.dy, F.dmask, F.w, F.seed_idx,
                           F.seed_bin, F.seed_thr, act, n_rem, F.trig, F.label, F.seed_size, F.queue);
        ++rounds;
    }
    LR_HIP(hipGetLastError());
    if (h_ctrl[kCtrlGen] > 0xF0000000u) {  // generation counter about to wrap: forget every tagged hash entry
        LR_HIP(hipMemsetAsync(B.slab_hash, 0, (size_t)B.n_slabs * B.slab_hash_cap * 32, s));
        LR_HIP(hipMemsetAsync(B.ctrl + kCtrlGen, 0, sizeof(uint32_t), s));
    }
    *rounds_out = rounds;
    if (tiers_out) {
        tiers_out[0] = h_ctrl[kCtrlBigTotal];
        tiers_out[1] = h_ctrl[kCtrlSlabTotal];
        tiers_out[2] = (h_ctrl[kCtrlStall] != 0u && h_ctrl[kCtrlNRemain] > 0u) ? h_ctrl[kCtrlNRemain] : 0u;
        tiers_out[3] = h_ctrl[kCtrlPhase];
        tiers_out[4] = h_ctrl[kCtrlWalked];
        tiers_out[5] = h_ctrl[kCtrlWalked + 1];
        tiers_out[6] = h_ctrl[kCtrlSteps];
        tiers_out[7] = h_ctrl[kCtrlSteps + 1];
        tiers_out[8] = h_ctrl[kCtrlBigLong];
        tiers_out[9] = h_ctrl[kCtrlMulti];
        tiers_out[10] = h_ctrl[kCtrlLogWalks];
        tiers_out[11] = h_ctrl[kCtrlLogGiveUp];
        tiers_out[12] = h_ctrl[kCtrlGiants];
        tiers_out[13] = h_ctrl[kCtrlGiantDone];
        tiers_out[14] = h_ctrl[kCtrlStaged];
        tiers_out[15] = h_ctrl[kCtrlQuietMiss];
    }
    return 0;
}

}  // namespace lramd
