// Stage 2 — seed selection and ordering.
//
// Reference: min_seed_value = mag.maxCoeff() * (1 - SEED_RATIO) (line_detector.cpp:209),
// find_peaks keeps (max5x5 == mag) && (mag > min_seed_value) and sorts by value, descending
// (filter.cpp:168-188).  The sort there is unstable; the canonical order of this build is
// (value desc, row asc, col asc), i.e. ascending order of the 64-bit key
//      ~bits(mag) : 32 | (row*w+col) : 29 | bin : 3
// which one radix sort delivers.  The candidate lists come per filter tile, so no global
// atomic counter is touched: per-tile pass counts, one exclusive scan, one ordered write.
#include <cstring>

#include <rocprim/rocprim.hpp>

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

// The sort runs on a fixed number of keys (`cap`, chosen by the host before it knows the seed count, so that no
// host round trip sits between the filter and the flood): the slots past the seeds are filled with the largest key.
__global__ __launch_bounds__(256) void seed_pad_kernel(uint64_t* __restrict__ keys, uint32_t cap,
                                                       const uint32_t* __restrict__ n_seeds) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i < cap && i >= *n_seeds) keys[i] = ~0ull;
}

__global__ __launch_bounds__(256) void seed_setup_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                         const float* __restrict__ dx, const float* __restrict__ dy,
                                                         BinTrig trig, float trace_tolerance,
                                                         int32_t* __restrict__ seed_idx, int32_t* __restrict__ seed_bin,
                                                         float* __restrict__ seed_thr) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = min(*n_ptr, cap);
    if (k >= n) return;
    const uint32_t lo = (uint32_t)keys[k];
    const uint32_t idx = lo >> 3;
    const int b = (int)(lo & 7u);
    // flood(): min_val = (1 - tolerance) * image(seed) with image = grad[seed_bin] (filter.cpp:112-113)
    const float v = directional(dx[idx], dy[idx], trig.st[b], trig.ct[b]);
    seed_idx[k] = (int32_t)idx;
    seed_bin[k] = b;
    seed_thr[k] = (1 - trace_tolerance) * v;
}

// ---- the seed order in two launches (frames of up to kOwnSortCap seeds) -------------------------------------------------
// rocPRIM sorts the 64-bit keys of a 4K frame in seven launches of 5-14 us, each far from filling the chip; with the
// pad launch before and the set-up launch after, the sort was nine launches and 60 us of a 1.6 ms frame.  Here: every
// workgroup sorts 4096 keys in LDS (slots past the seed count read as the largest key: no pad launch), then every key
// finds its place by counting the smaller keys in each of the other sorted blocks (keys are unique: they end in the pixel
// index) and writes the seed's record there directly (no sorted key array, no set-up launch).
constexpr uint32_t kSortBlock = 4096;
constexpr uint32_t kOwnSortBlocks = 32;
constexpr uint32_t kOwnSortCap = kSortBlock * kOwnSortBlocks;

// Bitonic sort of 4096 keys by 1024 threads, four consecutive keys per thread: a compare-exchange at distance 1 or 2 is
// inside a thread, at distance 4..128 inside a wavefront (shuffles, no barrier), and only the ten steps at distance
// >= 256 go through LDS with a barrier (33 us for the ten blocks of a 4K frame; all 78 steps through LDS: 37 us).
__device__ __forceinline__ uint64_t shfl_xor_u64(uint64_t v, int lane_mask) {
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)v, lane_mask);
    const uint32_t hi = (uint32_t)__shfl_xor((int)(uint32_t)(v >> 32), lane_mask);
    return ((uint64_t)hi << 32) | lo;
}
__global__ __launch_bounds__(1024) void seed_block_sort_kernel(uint64_t* __restrict__ keys, uint32_t cap,
                                                               const uint32_t* __restrict__ n_ptr) {
    __shared__ uint64_t sk[kSortBlock];
    const uint32_t n = min(*n_ptr, cap);
    const uint32_t base = blockIdx.x * kSortBlock;
    if (base >= n) return;  // nothing but padding (the ranking pass does not look at such a block)
    const uint32_t t = threadIdx.x, i0 = 4u * t;
    uint64_t e[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) e[r] = (base + i0 + r < n) ? keys[base + i0 + r] : ~0ull;
    // element i keeps the smaller of (own, partner) iff it is the lower one of an ascending pair or the upper one of a
    // descending pair
#define LR_KEEP(own, other, i, j, k) ((((((i) & (j)) == 0u) == (((i) & (k)) == 0u)) == ((other) < (own))) ? (other) : (own))
    for (uint32_t k = 2; k <= kSortBlock; k <<= 1) {
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
        if (base + i0 + r < cap) keys[base + i0 + r] = e[r];
}

__global__ __launch_bounds__(256) void seed_rank_setup_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ n_ptr,
                                                              uint32_t cap, const float* __restrict__ dx,
                                                              const float* __restrict__ dy, BinTrig trig, float trace_tolerance,
                                                              int32_t* __restrict__ seed_idx, int32_t* __restrict__ seed_bin,
                                                              float* __restrict__ seed_thr) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    const uint32_t n = min(*n_ptr, cap);
    const uint32_t n_blocks = (n + kSortBlock - 1) / kSortBlock;
    const uint32_t mine = g / kSortBlock;
    // The block sort writes (and pads with the largest key) only the slots below `cap`: a last block that `cap` cuts
    // short has nothing of this frame beyond it (zeros of a fresh allocation or an earlier frame's keys), and when `cap`
    // is the frame's pixel count there is no memory there at all.
    if (mine >= n_blocks || g >= cap) return;
    const uint64_t key = keys[g];
    if (key == ~0ull) return;  // padding of the last block
    uint32_t rank = g - mine * kSortBlock;
    for (uint32_t b = 0; b < n_blocks; ++b) {
        if (b == mine) continue;
        const uint64_t* __restrict__ blk = keys + (size_t)b * kSortBlock;
        uint32_t lo = 0, len = min(kSortBlock, cap - b * kSortBlock);  // count of keys below `key` in a sorted block
        while (len > 0) {
            const uint32_t half = len >> 1;
            if (blk[lo + half] < key) {
                lo += half + 1;
                len -= half + 1;
            } else {
                len = half;
            }
        }
        rank += lo;
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

// true when launch_seed_order handles this capacity itself (else: seed_pad + launch_seed_sort + launch_seed_setup)
bool seed_order_is_fused(uint32_t cap) {
    static const bool rocprim_only = std::getenv("LIBRECTIFY_SEED_SORT_ROCPRIM") != nullptr;  // (comparison knob)
    return !rocprim_only && cap <= kOwnSortCap;
}

int launch_seed_order(uint64_t* keys, const uint32_t* n_seeds, uint32_t cap, const float* dx, const float* dy, BinTrig trig,
                      float trace_tolerance, int32_t* seed_idx, int32_t* seed_bin, float* seed_thr, hipStream_t s) {
    if (cap == 0) return 0;
    const uint32_t nb = (cap + kSortBlock - 1) / kSortBlock;
    hipLaunchKernelGGL(seed_block_sort_kernel, dim3(nb), dim3(1024), 0, s, keys, cap, n_seeds);
    hipLaunchKernelGGL(seed_rank_setup_kernel, dim3((nb * kSortBlock + 255) / 256), dim3(256), 0, s, keys, n_seeds, cap, dx, dy,
                       trig, trace_tolerance, seed_idx, seed_bin, seed_thr);
    LR_HIP(hipGetLastError());
    return 0;
}

size_t seeds_temp_bytes(int n_tiles, size_t max_seeds) {
    (void)n_tiles;
    size_t b = 0;
    (void)rocprim::radix_sort_keys(nullptr, b, (uint64_t*)nullptr, (uint64_t*)nullptr, max_seeds, 0u, 64u);
    return b + 256;
}

int launch_seed_select(const uint64_t* cand, const uint32_t* cand_count, const uint32_t* tile_max, int n_tiles,
                       int cand_cap, float seed_keep_ratio, float* maxmag, uint32_t* tile_pass, uint32_t* tile_off, uint64_t* keys,
                       uint32_t key_cap, uint32_t* n_seeds, hipStream_t s) {
    const int blocks = (n_tiles + 3) / 4;
    hipLaunchKernelGGL(seed_count_kernel, dim3(blocks), dim3(256), 0, s, cand, cand_count, tile_max, n_tiles, cand_cap,
                       seed_keep_ratio, maxmag, tile_pass);
    hipLaunchKernelGGL(seed_scan_kernel, dim3(1), dim3(1024), 0, s, tile_pass, n_tiles, tile_off, n_seeds);
    hipLaunchKernelGGL(seed_write_kernel, dim3(blocks), dim3(256), 0, s, cand, cand_count, n_tiles, cand_cap, maxmag,
                       seed_keep_ratio, tile_pass, tile_off, keys, key_cap);
    if (!seed_order_is_fused(key_cap))
        hipLaunchKernelGGL(seed_pad_kernel, dim3((key_cap + 255) / 256), dim3(256), 0, s, keys, key_cap, n_seeds);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_seed_sort(uint64_t* keys_in, uint64_t* keys_out, uint32_t n, void* temp, size_t temp_bytes,
                     hipStream_t s) {
    if (n == 0) return 0;
    LR_HIP(rocprim::radix_sort_keys(temp, temp_bytes, keys_in, keys_out, (size_t)n, 0u, 64u, s));
    return 0;
}

int launch_seed_setup(const uint64_t* keys_sorted, const uint32_t* n_seeds, uint32_t cap, const float* dx, const float* dy,
                      BinTrig trig, float trace_tolerance, int32_t* seed_idx, int32_t* seed_bin, float* seed_thr,
                      hipStream_t s) {
    if (cap == 0) return 0;
    hipLaunchKernelGGL(seed_setup_kernel, dim3((cap + 255) / 256), dim3(256), 0, s, keys_sorted, n_seeds, cap, dx, dy, trig,
                       trace_tolerance, seed_idx, seed_bin, seed_thr);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
