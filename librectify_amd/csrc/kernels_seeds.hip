// Stage 2 — seed selection and ordering.
//
// Reference: min_seed_value = mag.maxCoeff() * (1 - SEED_RATIO) (line_detector.cpp:209),
// find_peaks keeps (max5x5 == mag) && (mag > min_seed_value) and sorts by value, descending
// (filter.cpp:168-188).  The sort there is unstable; the canonical order of this build is
// (value desc, row asc, col asc), i.e. ascending order of the 64-bit key
//      ~bits(mag) : 32 | (row*w+col) : 29 | bin : 3
// which one radix sort delivers.  The candidate lists come per filter tile, so no global
// atomic counter is touched: per-tile pass counts, one exclusive scan, one ordered write.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace lramd {
namespace {

// Seed threshold and per-band counts in one launch, their exclusive scan in a second (one workgroup); it used to be four
// (a one-workgroup max reduction, the counts, rocPRIM's scan with its own initialisation kernel), each some 4.5 us of
// launch floor in a chain that the flood waits for.
//  - min_seed_value needs the frame's largest magnitude (line_detector.cpp:209): every workgroup reduces ALL the per-band
//    maxima itself (20 KB from L2 for a 4K frame) instead of waiting for a kernel that does it once;
//  - one wavefront per band counts the band's candidates above the threshold (filter.cpp:168).
// (Folding the scan into the same launch -- the workgroup that finishes last does it -- was built and measured: the
// agent-scope fence every workgroup needs before it takes its ticket costs more than the launch it saves: 40 us for the
// fused kernel against 5 + 5.)
__global__ __launch_bounds__(256) void seed_count_kernel(const uint64_t* __restrict__ cand,
                                                         const uint32_t* __restrict__ cand_count,
                                                         const uint32_t* __restrict__ tile_max, int n_tiles, int cand_cap,
                                                         float keep_ratio, float* __restrict__ maxmag,
                                                         uint32_t* __restrict__ tile_pass) {
    __shared__ uint32_t s_red[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t m = 0;  // magnitudes are >= 0, so their bit patterns order like the floats
    // (eight loads in flight at a time, not one dependent L2 round trip per band maximum)
    for (int i0 = threadIdx.x; i0 < n_tiles; i0 += 256 * 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 256;
            v[k] = i < n_tiles ? tile_max[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) m = max(m, v[k]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    const float mx = __uint_as_float(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
    if (blockIdx.x == 0 && threadIdx.x == 0) *maxmag = mx;
    const float thr = mx * keep_ratio;
    const int tile = blockIdx.x * 4 + wave;
    if (tile >= n_tiles) return;
    const uint32_t n = cand_count[tile];
    const uint64_t* c = cand + (size_t)tile * cand_cap;
    uint32_t cnt = 0;
    for (uint32_t i = lane; i < n; i += 64) cnt += (__uint_as_float((uint32_t)(c[i] >> 32)) > thr) ? 1u : 0u;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off);
    if (lane == 0) tile_pass[tile] = cnt;
}

// Exclusive scan of the band counts by one workgroup of 1024 threads: every thread takes a contiguous run of the counts
// (loaded eight at a time, all in flight together), the runs' sums are scanned across the workgroup, and every thread
// writes its run's offsets; also writes the seed count.  Two barriers in all.
__global__ __launch_bounds__(1024) void seed_scan_kernel(const uint32_t* __restrict__ tile_pass, int n_tiles,
                                                         uint32_t* __restrict__ tile_off, uint32_t* __restrict__ n_seeds) {
    __shared__ uint32_t s_wave[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int K = (n_tiles + 1023) / 1024;
    const int i0 = (int)threadIdx.x * K, i1 = min(n_tiles, i0 + K);
    uint32_t sum = 0;
    for (int b0 = i0; b0 < i1; b0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = b0 + k < i1 ? tile_pass[b0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k) sum += v[k];
    }
    uint32_t inc = sum;  // inclusive scan of the runs' sums inside the wavefront ...
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    uint32_t before = 0, total = 0;  // ... and across the sixteen wavefronts
#pragma unroll
    for (int w2 = 0; w2 < 16; ++w2) {
        const uint32_t t = s_wave[w2];
        before += w2 < wave ? t : 0u;
        total += t;
    }
    uint32_t run = before + inc - sum;
    for (int b0 = i0; b0 < i1; b0 += 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = b0 + k < i1 ? tile_pass[b0 + k] : 0u;
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (b0 + k < i1) {
                tile_off[b0 + k] = run;
                run += v[k];
            }
    }
    if (threadIdx.x == 0) *n_seeds = total;
}

__global__ __launch_bounds__(256) void seed_write_kernel(const uint64_t* __restrict__ cand,
                                                         const uint32_t* __restrict__ cand_count, int n_tiles,
                                                         int cand_cap, const float* __restrict__ maxmag,
                                                         float keep_ratio, const uint32_t* __restrict__ tile_pass,
                                                         const uint32_t* __restrict__ tile_off,
                                                         uint64_t* __restrict__ keys, uint32_t cap) {
    const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (tile >= n_tiles) return;
    const float thr = *maxmag * keep_ratio;
    const uint32_t n = cand_count[tile];
    const uint64_t* c = cand + (size_t)tile * cand_cap;
    uint32_t base = tile_off[tile];
    for (uint32_t i0 = 0; i0 < n; i0 += 64) {
        const uint32_t i = i0 + lane;
        uint64_t k = 0;
        bool pass = false;
        if (i < n) {
            k = c[i];
            pass = __uint_as_float((uint32_t)(k >> 32)) > thr;
        }
        const uint64_t m = __ballot(pass);
        if (pass) {
            const uint32_t pos = base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            // beyond the sort's capacity: dropped, and the frame is repeated with a larger one (the count says so)
            if (pos < cap) keys[pos] = ((uint64_t)(~(uint32_t)(k >> 32)) << 32) | (k & 0xFFFFFFFFull);
        }
        base += (uint32_t)__popcll(m);
    }
}

// Round 5: threshold, counts, their scan and the ordered write in ONE launch (the three launches above
// stay for comparison: LIBRECTIFY_SEED_SELECT_FUSED=0).  At most 256 workgroups, each with a contiguous run of bands:
//   1. every workgroup reduces all the band maxima itself (as seed_count_kernel does) -> the threshold;
//   2. a wavefront per band counts the band's candidates above it, the counts stay in LDS; one wavefront scans them;
//   3. the workgroup PUBLISHES its total as one 64-bit word (frame tag : 32 | total : 32) with a relaxed agent-scope store --
//      the word is its own payload, so no fence is needed (the fence before a ticket is what made the "last workgroup does
//      the scan" version of round 4 cost 40 us) -- and adds up the words of the workgroups before it, waiting for each to
//      carry this frame's tag.  Workgroups are dispatched in index order and none waits for a later one: no deadlock; the
//      wait is bounded all the same (a stuck frame returns a wrong count rather than hang the device);
//   4. the bands' candidates above the threshold are written at (total before the workgroup) + (offset of the band).
// `status` holds at least gridDim.x words, zero when allocated; tags start at 1 and differ from frame to frame.
constexpr int kSelBands = 2048;  // bands per workgroup, at most (LDS)
__global__ __launch_bounds__(256) void seed_select_kernel(const uint64_t* __restrict__ cand, const uint32_t* __restrict__ cand_count,
                                                          const uint32_t* __restrict__ tile_max, int n_tiles, int cand_cap, int per_wg,
                                                          float keep_ratio, float* __restrict__ maxmag,
                                                          unsigned long long* __restrict__ status, uint32_t tag,
                                                          uint64_t* __restrict__ keys, uint32_t cap, uint32_t* __restrict__ n_seeds) {
    __shared__ uint32_t s_red[4], s_sum[4];
    __shared__ uint32_t s_cnt[kSelBands];
    __shared__ uint32_t s_total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t m = 0;
    for (int i0 = threadIdx.x; i0 < n_tiles; i0 += 256 * 8) {
        uint32_t v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = i0 + k * 256;
            v[k] = i < n_tiles ? tile_max[i] : 0u;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) m = max(m, v[k]);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
    if (lane == 0) s_red[wave] = m;
    __syncthreads();
    const float mx = __uint_as_float(max(max(s_red[0], s_red[1]), max(s_red[2], s_red[3])));
    if (blockIdx.x == 0 && threadIdx.x == 0) *maxmag = mx;
    const float thr = mx * keep_ratio;
    const int t0 = (int)blockIdx.x * per_wg, t1 = min(n_tiles, t0 + per_wg);
    for (int t = t0 + wave; t < t1; t += 4) {
        const uint32_t n = cand_count[t];
        const uint64_t* c = cand + (size_t)t * cand_cap;
        uint32_t cnt = 0;
        for (uint32_t i = lane; i < n; i += 64) cnt += (__uint_as_float((uint32_t)(c[i] >> 32)) > thr) ? 1u : 0u;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, off);
        if (lane == 0) s_cnt[t - t0] = cnt;
    }
    __syncthreads();
    const int nb = max(t1 - t0, 0);
    if (wave == 0) {  // exclusive scan of the bands' counts, 64 at a time; the workgroup's total is published at once
        uint32_t run = 0;
        for (int c0 = 0; c0 < nb; c0 += 64) {
            const uint32_t v = c0 + lane < nb ? s_cnt[c0 + lane] : 0u;
            uint32_t inc = v;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t o = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += o;
            }
            if (c0 + lane < nb) s_cnt[c0 + lane] = run + inc - v;
            run += (uint32_t)__shfl((int)inc, 63);
        }
        if (lane == 0) {
            s_total = run;
            __hip_atomic_store(&status[blockIdx.x], ((unsigned long long)tag << 32) | (unsigned long long)run, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // the totals of the workgroups before this one
    uint32_t before = 0;
    for (uint32_t j = threadIdx.x; j < blockIdx.x; j += 256) {
        unsigned long long w = 0ull;
        for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
            w = __hip_atomic_load(&status[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(w >> 32) == tag) break;
            __builtin_amdgcn_s_sleep(2);
        }
        before += (uint32_t)w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) before += (uint32_t)__shfl_xor((int)before, off);
    if (lane == 0) s_sum[wave] = before;
    __syncthreads();
    const uint32_t base0 = s_sum[0] + s_sum[1] + s_sum[2] + s_sum[3];
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *n_seeds = base0 + s_total;
    for (int t = t0 + wave; t < t1; t += 4) {
        const uint32_t n = cand_count[t];
        const uint64_t* c = cand + (size_t)t * cand_cap;
        uint32_t base = base0 + s_cnt[t - t0];
        for (uint32_t i0 = 0; i0 < n; i0 += 64) {
            const uint32_t i = i0 + lane;
            uint64_t k = 0;
            bool pass = false;
            if (i < n) {
                k = c[i];
                pass = __uint_as_float((uint32_t)(k >> 32)) > thr;
            }
            const uint64_t mm = __ballot(pass);
            if (pass) {
                const uint32_t pos = base + (uint32_t)__popcll(mm & ((1ull << lane) - 1ull));
                // beyond the sort's capacity: dropped, and the frame is repeated with a larger one (the count says so)
                if (pos < cap) keys[pos] = ((uint64_t)(~(uint32_t)(k >> 32)) << 32) | (k & 0xFFFFFFFFull);
            }
            base += (uint32_t)__popcll(mm);
        }
    }
}

// ---- the seed order: a sort of our own for every frame size ---------------------------------------------------------------
// The keys are unique (they end in the pixel index), so a key's place in the order is the number of smaller keys, and
// counting needs no exchange of data between workgroups:
//   1. every workgroup of 512 threads sorts a RUN of 2048 keys in LDS (bitonic; slots past the seed count read as the
//      largest key) -- 20 workgroups for the 40 000 seeds of a 4K frame (round 3: ten workgroups of 4096 keys, 33 us; round 4:
//      runs of 1024 keys and a merge round more);
//   2. while there are more than kFinalRuns runs: a merge round -- every key looks up its rank in the neighbouring run of
//      its pair and writes itself at (own index + that rank) of the merged run, twice as long (keys_a <-> keys_b);
//   3. every key counts the smaller keys of ALL other runs (branch-free lower bounds, eight runs side by side so that
//      eight loads are in flight per step instead of one) and writes the seed's record at that rank directly: no sorted key
//      array, no set-up launch.
// A 4K frame is 1 + 1 launches (8 + 20 us), an 8192 x 8192 frame with 330 000 seeds 1 + 5 + 1.  rocPRIM's radix sort, which
// round 3 still used above 131 072 keys (nine launches with its pad and set-up kernels), is gone from the path.
// The host knows only the CAPACITY the frame runs with (`cap`); the seed count n stays on the device: grids cover the
// capacity, and everything past n is skipped by count, never by a padding value in memory.
constexpr uint32_t kRun0 = 2048;      // keys per sorted run of step 1 (round 5: 2048 by 512 threads -- a merge round less for a 4K frame, 19 us instead of 15 + 5 and a launch)
constexpr uint32_t kRunThreads = kRun0 / 4;
constexpr uint32_t kFinalRuns = 16;   // step 3 takes over when the capacity is this many runs or fewer (its cost grows with their number)

__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int lane_mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, lane_mask);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), lane_mask);
    return ((uint64_t)hi << 32) | lo;
}
// Bitonic sort of kRun0 keys by kRun0 / 4 threads, four consecutive keys per thread: a compare-exchange at distance 1 or 2 is
// inside a thread, at distance 4..128 inside a wavefront (shuffles, no barrier), and only the steps at distance
// >= 256 (six of the 66 for 2048 keys) go through LDS with a barrier.
__global__ __launch_bounds__(kRunThreads) void seed_run_sort_kernel(uint64_t* __restrict__ keys, uint32_t cap,
                                                            const uint32_t* __restrict__ n_ptr) {
    __shared__ uint64_t sk[kRun0];
    const uint32_t n = min(*n_ptr, cap);
    const uint32_t base = blockIdx.x * kRun0;
    if (base >= n) return;  // nothing of this frame here
    const uint32_t t = threadIdx.x, i0 = 4u * t;
    uint64_t e[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = (base + i0 + r < n) ? keys[base + i0 + r] : ~0ull;
    // element i keeps the smaller of (own, partner) iff it is the lower one of an ascending pair or the upper one of a
    // descending pair
#define LR_KEEP(own, other, i, j, k) ((((((i) & (j)) == 0u) == (((i) & (k)) == 0u)) == ((other) < (own))) ? (other) : (own))
    for (uint32_t k = 2; k <= kRun0; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            if (j >= 256u) {
#pragma unroll
                for (int r = 0; r < 4; ++r) sk[i0 + r] = e[r];
                __syncthreads();
                uint64_t o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = sk[(i0 + r) ^ j];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) e[r] = LR_KEEP(e[r], o[r], i0 + r, j, k);
            } else if (j >= 4u) {
                const int lm = (int)(j >> 2);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const uint64_t o = shfl_xor_u64(e[r], lm);
                    e[r] = LR_KEEP(e[r], o, i0 + r, j, k);
                }
            } else {
                const bool two = j == 2u;  // (no run-time index into the register array)
                const uint64_t o0 = two ? e[2] : e[1], o1 = two ? e[3] : e[0], o2 = two ? e[0] : e[3], o3 = two ? e[1] : e[2];
                e[0] = LR_KEEP(e[0], o0, i0 + 0u, j, k);
                e[1] = LR_KEEP(e[1], o1, i0 + 1u, j, k);
                e[2] = LR_KEEP(e[2], o2, i0 + 2u, j, k);
                e[3] = LR_KEEP(e[3], o3, i0 + 3u, j, k);
            }
        }
    }
#undef LR_KEEP
#pragma unroll
    for (int r = 0; r < 4; ++r)
        if (base + i0 + r < n) keys[base + i0 + r] = e[r];  // (the padding stays in the registers)
}

// keys of run q (length L, a power of two; `valid` of them belong to the frame) that are smaller than `key`: a lower bound
// in exactly log2 L steps without a data-dependent branch (a position past the valid keys counts as "not smaller")
__device__ __forceinline__ uint32_t count_below(const uint64_t* __restrict__ run, uint32_t L, uint32_t valid, uint64_t key) {
    uint32_t lo = 0;
    for (uint32_t sstep = L >> 1; sstep > 0; sstep >>= 1) {
        const uint32_t pos = lo + sstep - 1u;
        const uint64_t v = pos < valid ? run[pos] : ~0ull;
        lo += v < key ? sstep : 0u;
    }
    // (lo counts the keys at positions 0 .. L-2 that are smaller; the last position)
    return lo + ((lo == L - 1u && L - 1u < valid && run[L - 1u] < key) ? 1u : 0u);
}

// Step 2: runs of length L merged in pairs into runs of 2 L.
__global__ __launch_bounds__(256) void seed_merge_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, uint32_t L,
                                                         uint32_t cap, const uint32_t* __restrict__ n_ptr) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = min(*n_ptr, cap);
    if (g >= n) return;
    const uint32_t r = g / L, i = g - r * L, q = r ^ 1u;
    const uint64_t key = in[g];
    const uint32_t qbase = q * L;
    const uint32_t valid = qbase < n ? min(L, n - qbase) : 0u;
    const uint32_t below = valid ? count_below(in + qbase, L, valid, key) : 0u;
    out[(size_t)(r & ~1u) * L + i + below] = key;
}

// Step 3: ranks among all runs, and the seed records at their ranks.
__global__ __launch_bounds__(256) void seed_rank_setup_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr,
                                                              uint32_t cap, uint32_t L, const float* __restrict__ dx,
                                                              const float* __restrict__ dy, BinTrig trig, float trace_tolerance,
                                                              int32_t* __restrict__ seed_idx, int32_t* __restrict__ seed_bin,
                                                              float* __restrict__ seed_thr) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = min(*n_ptr, cap);
    if (g >= n) return;
    const uint32_t n_runs = (n + L - 1u) / L;
    const uint32_t mine = g / L;
    const uint64_t key = keys[g];
    uint32_t rank = g - mine * L;
    // eight runs side by side: the eight lower bounds advance in lock step, eight loads in flight per step
    for (uint32_t q0 = 0; q0 < n_runs; q0 += 8u) {
        uint32_t lo[8], valid[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const uint32_t q = q0 + (uint32_t)j;
            lo[j] = 0u;
            valid[j] = (q < n_runs && q != mine) ? min(L, n - q * L) : 0u;
        }
        for (uint32_t sstep = L >> 1; sstep > 0; sstep >>= 1) {
            uint64_t v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const uint32_t pos = lo[j] + sstep - 1u;
                v[j] = pos < valid[j] ? keys[(size_t)(q0 + (uint32_t)j) * L + pos] : ~0ull;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) lo[j] += v[j] < key ? sstep : 0u;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            // (the last position of a full run: see count_below)
            if (lo[j] == L - 1u && L - 1u < valid[j] && keys[(size_t)(q0 + (uint32_t)j) * L + L - 1u] < key) lo[j] += 1u;
            rank += lo[j];
        }
    }
    const uint32_t low = (uint32_t)key;
    const uint32_t idx = low >> 3;
    const int bin = (int)(low & 7u);
    // flood(): min_val = (1 - tolerance) * image(seed) with image = grad[seed_bin] (filter.cpp:112-113)
    const float v = directional(dx[idx], dy[idx], trig.st[bin], trig.ct[bin]);
    seed_idx[rank] = (int32_t)idx;
    seed_bin[rank] = bin;
    seed_thr[rank] = (1 - trace_tolerance) * v;
}

}  // namespace

// keys / keys_alt: two buffers of at least `cap` keys; the unsorted keys are in `keys`
int launch_seed_order(uint64_t* keys, uint64_t* keys_alt, const uint32_t* n_seeds, uint32_t cap, const float* dx, const float* dy,
                      BinTrig trig, float trace_tolerance, int32_t* seed_idx, int32_t* seed_bin, float* seed_thr, hipStream_t s) {
    if (cap == 0) return 0;
    const uint32_t runs0 = (cap + kRun0 - 1) / kRun0;
    hipLaunchKernelGGL(seed_run_sort_kernel, dim3(runs0), dim3(kRunThreads), 0, s, keys, cap, n_seeds);
    uint32_t L = kRun0, runs = runs0;
    uint64_t* cur = keys;
    uint64_t* nxt = keys_alt;
    // (step 3 costs (keys) x (runs) x log2 L loads, a merge round (keys) x log2 L and a launch: merge while step 3 would be
    // the larger part -- an 8192 x 8192 frame with room for 490 000 seeds goes down to 8 runs in six rounds)
    // (never past one run: with a capacity above 4 Mi seeds -- a frame of more than 134 Mpix, a dense frame after adapt_seed_cap,
    // lr_set_seed_capacity -- the second condition alone stayed true for ever: the host enqueued merges until L wrapped)
    while (runs > 1u && (runs > kFinalRuns || (uint64_t)cap * runs > ((uint64_t)4 << 20))) {
        hipLaunchKernelGGL(seed_merge_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, cur, nxt, L, cap, n_seeds);
        std::swap(cur, nxt);
        L *= 2u;
        runs = (runs + 1u) / 2u;
    }
    hipLaunchKernelGGL(seed_rank_setup_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, cur, n_seeds, cap, L, dx, dy, trig,
                       trace_tolerance, seed_idx, seed_bin, seed_thr);
    LR_HIP(hipGetLastError());
    return 0;
}

size_t seeds_temp_bytes(int n_tiles, size_t max_seeds) {
    (void)n_tiles;
    (void)max_seeds;
    return 256;  // (the seed order needs no scratch beyond the two key buffers)
}

int launch_seed_select(const uint64_t* cand, const uint32_t* cand_count, const uint32_t* tile_max, int n_tiles,
                       int cand_cap, float seed_keep_ratio, float* maxmag, uint32_t* tile_pass, uint32_t* tile_off, uint64_t* keys,
                       uint32_t key_cap, uint32_t* n_seeds, uint32_t frame_tag, hipStream_t s) {
    static const bool fused = !(std::getenv("LIBRECTIFY_SEED_SELECT_FUSED") && std::atoi(std::getenv("LIBRECTIFY_SEED_SELECT_FUSED")) == 0);
    // (the status words are tile_off's: n_tiles words, zero when allocated -- two words a workgroup, at most n_tiles / 4 workgroups)
    if (fused && frame_tag != 0u && n_tiles > 0) {
        int wgs = std::min(256, (n_tiles + 3) / 4);
        int per_wg = (n_tiles + wgs - 1) / wgs;
        if (per_wg > kSelBands) {
            per_wg = kSelBands;
            wgs = (n_tiles + per_wg - 1) / per_wg;
        }
        wgs = (n_tiles + per_wg - 1) / per_wg;  // (no workgroup without a band: the last one writes the count)
        hipLaunchKernelGGL(seed_select_kernel, dim3(wgs), dim3(256), 0, s, cand, cand_count, tile_max, n_tiles, cand_cap, per_wg,
                           seed_keep_ratio, maxmag, reinterpret_cast<unsigned long long*>(tile_off), frame_tag, keys, key_cap, n_seeds);
        LR_HIP(hipGetLastError());
        return 0;
    }
    const int blocks = (n_tiles + 3) / 4;
    hipLaunchKernelGGL(seed_count_kernel, dim3(blocks), dim3(256), 0, s, cand, cand_count, tile_max, n_tiles, cand_cap,
                       seed_keep_ratio, maxmag, tile_pass);
    hipLaunchKernelGGL(seed_scan_kernel, dim3(1), dim3(1024), 0, s, tile_pass, n_tiles, tile_off, n_seeds);
    hipLaunchKernelGGL(seed_write_kernel, dim3(blocks), dim3(256), 0, s, cand, cand_count, n_tiles, cand_cap, maxmag,
                       seed_keep_ratio, tile_pass, tile_off, keys, key_cap);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
