// Stage 4 — components from the label image and the weighted-PCA line fit
// (reference line_detector.cpp:66-89,111, geometry.cpp:20-61).
//
// Components are the floods with more than COMPONENT_MIN_SIZE pixels, in seed order.  Their
// pixel lists are rebuilt from the label image (scatter by label, then a sort of every list
// by pixel index), which puts every component in the canonical row-major order regardless of
// how the flood discovered it.  One wavefront fits one component with the canonical
// reduction tree T(): lane-strided sequential partial sums, then an xor butterfly.
//
// No size in this stage is known to the host when it enqueues the kernels (seed, component and
// pixel counts stay on the device until the frame's single synchronisation): launches cover the
// capacity the seed sort ran with and every kernel reads the real counts from memory.
#include <algorithm>
#include <cstring>

#include <atomic>

#include "common.h"

namespace lramd {
namespace {

constexpr uint32_t kNoComp = 0xFFFFFFFFu;

// Rank and pixel offset of every kept flood, in seed order: a two-level scan.  Workgroup b owns seeds
// [b * kOffChunk, (b + 1) * kOffChunk), eight consecutive seeds per thread.  First kernel: per-chunk totals.
// Second kernel: every workgroup adds up the totals of the chunks before it and scans its own chunk.
constexpr uint32_t kSortLds = 1024;  // longest pixel list sorted by a 256-thread workgroup (4 KB of LDS); longer ones take 1024 threads
constexpr uint32_t kSortLdsBig = 16384;  // ... in 64 KB of LDS
// Lists beyond that ("huge": a flood of a smooth region, a ring of a noiseless gradient -- hundreds of thousands of pixels)
// were sorted in place in global memory by ONE workgroup: 7.7 ms for a 140 000-pixel flood, 53 ms for sixteen rings at
// 1080p.  The keys are distinct pixel indices, so a run of 2^14 consecutive indices holds at most 2^14 of them whatever
// the component looks like: the pixels of a huge component are dealt into buckets by index >> 14 (counted first, so that
// every bucket has its place in the component's slice), and the buckets are sorted in LDS by as many workgroups as there
// are.  Words of `n_large` (the frame's d_counts + 4): [0] lists of 65..1024 pixels, [1] longer ones, [2] huge
// components, [3] their non-empty buckets, [6] workgroups of huge_count_kernel that have finished.
constexpr uint32_t kHugeShift = 14;
constexpr uint32_t kHugeFlag = 0x80000000u;  // cursor[component]: the component's number among the huge ones, flagged
constexpr int kOffPer = 8;
constexpr uint32_t kOffChunk = 256 * kOffPer;

__device__ __forceinline__ void chunk_counts(const int32_t* __restrict__ seed_size, uint32_t n_seeds, int min_size,
                                             uint32_t k0, int (&sz)[kOffPer], uint32_t& c, uint32_t& p) {
    c = p = 0;
#pragma unroll
    for (int j = 0; j < kOffPer; ++j) {
        sz[j] = (k0 + j < n_seeds) ? seed_size[k0 + j] : 0;
        if (sz[j] > min_size) {
            c += 1u;
            p += (uint32_t)sz[j];
        }
    }
}

// Component numbers and pixel offsets of the kept floods: an exclusive scan over the seeds in ONE launch (round 5; until then a
// launch for the chunks' totals went first).  Every workgroup counts its chunk of 2048 seeds, PUBLISHES the two totals as
// tagged 64-bit words (frame tag : 32 | value : 32; relaxed agent-scope stores -- the word is its own payload, no fence) and
// adds up the words of the chunks before it, waiting for each to carry this frame's tag (workgroups are dispatched in index
// order and none waits for a later one; the wait is bounded all the same).  Chunk 0 clears the list counters (n_large) before
// it publishes -- with release order, the one fence of the launch -- and nobody touches them before having seen its words.
// `status`: two words per chunk, zero when allocated; tags start at 1 and differ from frame to frame.
__global__ __launch_bounds__(256) void component_offsets_kernel(const int32_t* __restrict__ seed_size,
                                                                const uint32_t* __restrict__ n_ptr, uint32_t cap,
                                                                int min_size, unsigned long long* __restrict__ status, uint32_t tag,
                                                                uint32_t* __restrict__ comp_rank,
                                                                uint32_t* __restrict__ comp_seed,
                                                                uint32_t* __restrict__ comp_off,
                                                                uint32_t* __restrict__ totals,
                                                                uint32_t* __restrict__ large_list,
                                                                uint32_t large_cap, uint32_t* __restrict__ n_large,
                                                                uint32_t* __restrict__ cursor, uint32_t* __restrict__ huge_list,
                                                                uint32_t huge_max) {
    __shared__ uint32_t s_c[4], s_p[4], s_cc[4], s_cp[4];
    const uint32_t n_seeds = min(*n_ptr, cap);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint32_t k0 = blockIdx.x * kOffChunk + (uint32_t)tid * kOffPer;
    int sz[kOffPer];
    uint32_t c, p;
    chunk_counts(seed_size, n_seeds, min_size, k0, sz, c, p);
    uint32_t ic = c, ip = p;  // inclusive scan inside the wave
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t tc = (uint32_t)__shfl_up((int)ic, off);
        const uint32_t tp = (uint32_t)__shfl_up((int)ip, off);
        if (lane >= off) {
            ic += tc;
            ip += tp;
        }
    }
    if (lane == 63) {
        s_c[wv] = ic;
        s_p[wv] = ip;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long t = (unsigned long long)tag << 32;
        const unsigned long long wc = t | (unsigned long long)(s_c[0] + s_c[1] + s_c[2] + s_c[3]);
        const unsigned long long wp = t | (unsigned long long)(s_p[0] + s_p[1] + s_p[2] + s_p[3]);
        if (blockIdx.x == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) __hip_atomic_store(&n_large[i], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&status[0], wc, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&status[1], wp, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            __hip_atomic_store(&status[2u * blockIdx.x], wc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&status[2u * blockIdx.x + 1u], wp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // totals of the chunks before this one
    uint32_t cc = 0, cp = 0;
    for (uint32_t i = (uint32_t)tid; i < blockIdx.x; i += 256) {
        unsigned long long wc = 0ull, wp = 0ull;
        for (uint32_t spins = 0; spins < (1u << 22); ++spins) {
            wc = __hip_atomic_load(&status[2u * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            wp = __hip_atomic_load(&status[2u * i + 1u], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((uint32_t)(wc >> 32) == tag && (uint32_t)(wp >> 32) == tag) break;
            __builtin_amdgcn_s_sleep(2);
        }
        cc += (uint32_t)wc;
        cp += (uint32_t)wp;
    }
    // (thread 0 of every later workgroup has looked at chunk 0's words: the barrier below puts the whole workgroup's atomics on
    // n_large behind chunk 0's clearing of it)
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cc += (uint32_t)__shfl_xor((int)cc, off);
        cp += (uint32_t)__shfl_xor((int)cp, off);
    }
    if (lane == 0) {
        s_cc[wv] = cc;
        s_cp[wv] = cp;
    }
    __syncthreads();
    uint32_t rank = s_cc[0] + s_cc[1] + s_cc[2] + s_cc[3] + ic - c;
    uint32_t off_px = s_cp[0] + s_cp[1] + s_cp[2] + s_cp[3] + ip - p;
    for (int i = 0; i < wv; ++i) {
        rank += s_c[i];
        off_px += s_p[i];
    }
#pragma unroll
    for (int j = 0; j < kOffPer; ++j) {
        const uint32_t k = k0 + j;
        if (k < n_seeds) {
            const bool keep = sz[j] > min_size;
            comp_rank[k] = keep ? rank : kNoComp;
            if (keep) {
                comp_seed[rank] = k;
                comp_off[rank] = off_px;
                // lists of more than 64 pixels are sorted by a workgroup each (component_sort_large_kernel): in LDS up to
                // 1024 pixels (list filled from the front), through global memory beyond (filled from the back)
                if (sz[j] > (int)kSortLds) large_list[large_cap - 1u - atomicAdd(n_large + 1, 1u)] = rank;
                else if (sz[j] > 64) large_list[atomicAdd(n_large, 1u)] = rank;
                // the scatter pass's cursor of this component: zero, or -- flagged -- its number among the huge ones (the
                // scatter pass finds the buckets through it).  Every component's word is written here: no memset launch.
                uint32_t cur = 0u;
                if (sz[j] > (int)kSortLdsBig && huge_max != 0u) {  // (beyond huge_max -- never: the old way, component_sort_big_kernel<true>)
                    const uint32_t hi = atomicAdd(n_large + 2, 1u);
                    if (hi < huge_max) {
                        cur = hi | kHugeFlag;
                        huge_list[hi] = rank;
                    }
                }
                cursor[rank] = cur;
                rank += 1u;
                off_px += (uint32_t)sz[j];
            }
        }
    }
    if (blockIdx.x == gridDim.x - 1 && tid == 255) {  // the last thread of the last chunk ends on the grand totals
        totals[0] = rank;
        totals[1] = off_px;
        comp_off[rank] = off_px;
    }
}

// Pixels of kept floods go to their component's slice of `px` (any order: the segmented sort follows).  Lanes of
// a wavefront that hold the same component (neighbours along a row mostly do) take their slots with ONE atomic:
// per-pixel atomics on the cursor of a 3000-pixel component would queue up behind each other in L2.
__global__ __launch_bounds__(256) void component_scatter_kernel(const uint32_t* __restrict__ label, size_t npix,
                                                                const uint32_t* __restrict__ comp_rank,
                                                                const uint32_t* __restrict__ comp_off,
                                                                uint32_t* __restrict__ cursor,
                                                                uint32_t* __restrict__ px, uint32_t* __restrict__ huge_tab,
                                                                uint32_t nb) {
    const int lane = threadIdx.x & 63;
    const size_t step = (size_t)gridDim.x * 256;
    const size_t n_pad = (npix + 255) / 256 * 256;  // whole wavefronts take part in the ballots
    // (four of a thread's pixels at a time: their labels, then the ranks those point to, are in flight together -- the pass is
    // two dependent loads and an atomic with a returned value per wavefront and component, and nearly all latency)
    constexpr int kU = 4;
    for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n_pad; i0 += step * kU) {
        uint32_t l[kU], r[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t i = i0 + step * (size_t)u;
            l[u] = i < npix ? label[i] : kLabelFree;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) r[u] = l[u] != kLabelFree ? comp_rank[l[u]] : kNoComp;
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const size_t i = i0 + step * (size_t)u;
            if (i >= n_pad) break;  // (uniform: i0 and step are multiples of the wavefront within a workgroup's stride)
            uint64_t todo = __ballot(r[u] != kNoComp);
            while (todo != 0ull) {
                const int leader = __builtin_ctzll(todo);
                const uint32_t rl = (uint32_t)__builtin_amdgcn_readlane((int)r[u], leader);
                const uint64_t same = __ballot(r[u] == rl) & todo;
                uint32_t base = 0;
                if (lane == leader) {
                    const uint32_t o0 = comp_off[rl], o1 = comp_off[rl + 1u];
                    uint32_t hi = 0u;
                    if (o1 - o0 > kSortLdsBig) hi = cursor[rl];  // (a huge component's word is its number, never counted on)
                    // (a wavefront's 64 consecutive pixels lie in one bucket: 2^14 is a multiple of 64)
                    if (hi & kHugeFlag) base = atomicAdd(&huge_tab[(size_t)(hi & ~kHugeFlag) * nb + (uint32_t)(i >> kHugeShift)], (uint32_t)__popcll(same));
                    else base = o0 + atomicAdd(&cursor[rl], (uint32_t)__popcll(same));
                }
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                if ((same >> lane) & 1ull) px[base + (uint32_t)__popcll(same & ((1ull << lane) - 1ull))] = (uint32_t)i;
                todo &= ~same;
            }
        }
    }
}

__device__ inline float wave_tree(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

// Closed-form symmetric 2x2 eigen-decomposition in double (sqrt and divide only): unit major
// axis (row, col), sign chosen so that row-component >= col-component (the orientation the
// reference's Eigen solver produces on all 848 rows of doc/image.jpg_warp_lines.csv).
__device__ inline void major_axis_2x2(float a_, float b_, float c_, float& d_r, float& d_c) {
    const double a = a_, b = b_, c = c_;
    const double hd = (a - c) * 0.5;
    const double rad = sqrt(hd * hd + b * b);
    const double lmax = (a + c) * 0.5 + rad;
    double vr, vc;
    if (a >= c) {
        vr = lmax - c;
        vc = b;
    } else {
        vr = b;
        vc = lmax - a;
    }
    const double nn = sqrt(vr * vr + vc * vc);
    if (nn > 0.0) {
        vr = vr / nn;
        vc = vc / nn;
    } else {
        vr = 1.0;
        vc = 0.0;
    }
    float fr = (float)vr, fc = (float)vc;
    if (fr < fc || (fr == fc && fr < 0.0f)) {
        fr = -fr;
        fc = -fc;
    }
    d_r = fr;
    d_c = fc;
}

// fit_line_parameters (geometry.cpp:20-61) for one component per wavefront.
// ---- per-component sort of the pixel lists ---------------------------------------------------------------
// Lists of up to 64 pixels (nearly all of them: the mean flood has some 25 pixels) are not sorted by a kernel of their
// own any more: fit_kernel takes them unsorted, one pixel per lane, ranks them (rank = number of smaller keys: the keys are
// distinct; 64 scalar broadcasts) and permutes them into order across the wavefront, all in registers.
// Longer lists: one workgroup each, bitonic network on the list padded to a power of two.  Three classes, told apart
// when the offsets are computed (component_offsets_kernel): up to 1024 pixels in 4 KB of LDS by 256 threads (many
// workgroups; entries at the front of large_list); up to 16384 pixels in 64 KB of LDS by 1024 threads (a 4096-pixel list
// took the 256-thread kernel 40 us -- every pass of the network eight trips through its loop -- and was what a 4K frame's
// sort stage waited for); beyond that in
// place in global memory by a single workgroup (a flood of more than 16384 pixels: rare, and slow here -- every pass
// is a round trip to L2).  The last two share the entries at the back of large_list.
template <int kThreads, class Keys>
__device__ __forceinline__ void bitonic_sort(Keys& key, uint32_t P) {
    for (uint32_t k = 2; k <= P; k <<= 1)
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = threadIdx.x; t < P / 2; t += kThreads) {
                const uint32_t i = 2 * t - (t & (j - 1));  // index with bit j clear
                const uint32_t a = key[i], b = key[i + j];
                const bool up = (i & k) == 0;
                if ((a > b) == up) {
                    key[i] = b;
                    key[i + j] = a;
                }
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(256) void component_sort_large_kernel(const uint32_t* __restrict__ px_in,
                                                                   uint32_t* __restrict__ px_out,
                                                                   const uint32_t* __restrict__ comp_off,
                                                                   const uint32_t* __restrict__ large_list,
                                                                   const uint32_t* __restrict__ n_large) {
    __shared__ uint32_t s_key[kSortLds];
    const uint32_t n_list = n_large[0];
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t comp = large_list[li];
        const uint32_t off = comp_off[comp];
        const uint32_t n = comp_off[comp + 1] - off;
        uint32_t P = 128;
        while (P < n) P <<= 1;
        for (uint32_t i = threadIdx.x; i < P; i += 256) s_key[i] = i < n ? px_in[off + i] : 0xFFFFFFFFu;
        __syncthreads();
        bitonic_sort<256>(s_key, P);
        for (uint32_t i = threadIdx.x; i < n; i += 256) px_out[off + i] = s_key[i];
        __syncthreads();
    }
}

// kGlobal = false: lists of 4097..16384 pixels, in LDS; true: longer ones, in `scratch` (>= 2 x the frame's pixels;
// launched with ONE workgroup, which takes them one after the other)
template <bool kGlobal>
__global__ __launch_bounds__(1024) void component_sort_big_kernel(const uint32_t* __restrict__ px_in,
                                                                  uint32_t* __restrict__ px_out,
                                                                  const uint32_t* __restrict__ comp_off,
                                                                  const uint32_t* __restrict__ large_list,
                                                                  uint32_t large_cap,
                                                                  const uint32_t* __restrict__ n_large,
                                                                  uint32_t* __restrict__ scratch,
                                                                  const uint32_t* __restrict__ cursor) {
    __shared__ uint32_t s_key[kGlobal ? 1 : kSortLdsBig];
    const uint32_t n_list = n_large[1];
    for (uint32_t li = blockIdx.x; li < n_list; li += gridDim.x) {
        const uint32_t comp = large_list[large_cap - 1u - li];
        const uint32_t off = comp_off[comp];
        const uint32_t n = comp_off[comp + 1] - off;
        if ((n > kSortLdsBig) != kGlobal) continue;  // the other launch's
        if (kGlobal && cursor != nullptr && (cursor[comp] & kHugeFlag)) continue;  // dealt into buckets: huge_sort_kernel
        uint32_t P = 2048;
        while (P < n) P <<= 1;
        if (!kGlobal) {
            for (uint32_t i = threadIdx.x; i < P; i += 1024) s_key[i] = i < n ? px_in[off + i] : 0xFFFFFFFFu;
            __syncthreads();
            bitonic_sort<1024>(s_key, P);
            for (uint32_t i = threadIdx.x; i < n; i += 1024) px_out[off + i] = s_key[i];
            __syncthreads();
        } else {
            for (uint32_t i = threadIdx.x; i < P; i += 1024) scratch[i] = i < n ? px_in[off + i] : 0xFFFFFFFFu;
            __syncthreads();
            bitonic_sort<1024>(scratch, P);
            for (uint32_t i = threadIdx.x; i < n; i += 1024) px_out[off + i] = scratch[i];
            __syncthreads();
        }
    }
}

// Pixels of huge components per bucket, then (the last workgroup to finish) every component's buckets in place: a
// bucket's word becomes the position of its first pixel in the sorted list -- the scatter pass takes the pixels' places from
// there -- and every non-empty bucket is a job of huge_sort_kernel.  Leaves at once on a frame without huge components.
__global__ __launch_bounds__(256) void huge_count_kernel(const uint32_t* __restrict__ label, size_t npix,
                                                         const uint32_t* __restrict__ comp_rank,
                                                         const uint32_t* __restrict__ comp_off,
                                                         const uint32_t* __restrict__ cursor, uint32_t* __restrict__ huge_tab,
                                                         uint32_t nb, const uint32_t* __restrict__ huge_list, uint32_t huge_max,
                                                         uint4* __restrict__ jobs, uint32_t* __restrict__ n_large) {
    const uint32_t n_huge = min(n_large[2], huge_max);
    if (n_huge == 0u) return;
    const int lane = threadIdx.x & 63;
    const size_t step = (size_t)gridDim.x * 256;
    const size_t n_pad = (npix + 255) / 256 * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n_pad; i += step) {
        const uint32_t l = i < npix ? label[i] : kLabelFree;
        const uint32_t r = l != kLabelFree ? comp_rank[l] : kNoComp;
        uint32_t hi = 0u;
        if (r != kNoComp && comp_off[r + 1u] - comp_off[r] > kSortLdsBig) hi = cursor[r];
        uint64_t todo = __ballot((hi & kHugeFlag) != 0u);
        while (todo != 0ull) {
            const int leader = __builtin_ctzll(todo);
            const uint32_t hl = (uint32_t)__builtin_amdgcn_readlane((int)hi, leader);
            const uint64_t same = __ballot(hi == hl) & todo;
            if (lane == leader) atomicAdd(&huge_tab[(size_t)(hl & ~kHugeFlag) * nb + (uint32_t)(i >> kHugeShift)], (uint32_t)__popcll(same));
            todo &= ~same;
        }
    }
    __shared__ uint32_t s_last;
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(n_large + 6, 1u) == gridDim.x - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last == 0u) return;
    __threadfence();
    if (threadIdx.x == 0) n_large[6] = 0u;
    // a wavefront a component, 64 buckets at a time
    for (uint32_t h = threadIdx.x >> 6; h < n_huge; h += 4u) {
        const uint32_t rank = huge_list[h];
        uint32_t run = comp_off[rank];
        for (uint32_t b0 = 0; b0 < nb; b0 += 64u) {
            const uint32_t b = b0 + (uint32_t)lane;
            uint32_t* e = &huge_tab[(size_t)h * nb + b];
            const uint32_t cnt = b < nb ? __hip_atomic_load(e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            uint32_t inc = cnt;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t t = (uint32_t)__shfl_up((int)inc, off);
                if (lane >= off) inc += t;
            }
            const uint32_t start = run + inc - cnt;
            if (cnt != 0u) {
                *e = start;
                jobs[atomicAdd(n_large + 3, 1u)] = make_uint4(start, cnt, h * nb + b, 0u);
            }
            run += (uint32_t)__shfl((int)inc, 63);
        }
    }
}

// A bucket of a huge component (at most 2^14 pixels: the indices of a run of 2^14 consecutive pixels) per workgroup at a
// time, sorted in LDS; the bucket's word in the table goes back to zero for the next frame.
__global__ __launch_bounds__(1024) void huge_sort_kernel(const uint32_t* __restrict__ px_in, uint32_t* __restrict__ px_out,
                                                         const uint4* __restrict__ jobs, uint32_t* __restrict__ huge_tab,
                                                         const uint32_t* __restrict__ n_large) {
    __shared__ uint32_t s_key[kSortLdsBig];
    const uint32_t n_jobs = n_large[3];
    for (uint32_t j = blockIdx.x; j < n_jobs; j += gridDim.x) {
        const uint4 job = jobs[j];
        const uint32_t off = job.x, n = job.y;
        uint32_t P = 64;
        while (P < n) P <<= 1;
        for (uint32_t i = threadIdx.x; i < P; i += 1024) s_key[i] = i < n ? px_in[off + i] : 0xFFFFFFFFu;
        __syncthreads();
        bitonic_sort<1024>(s_key, P);
        for (uint32_t i = threadIdx.x; i < n; i += 1024) px_out[off + i] = s_key[i];
        if (threadIdx.x == 0) huge_tab[job.z] = 0u;
        __syncthreads();
    }
}

// Components of more than 64 pixels: a lane takes the pixels lane, lane + 64, ... IN THAT ORDER (the canonical order of the
// sums), kU of them in flight at a time: the pixel list first, then the gathers it points to -- one round trip per kU pixels
// instead of one per pixel.  The launch lasts as long as its longest component: 3 700 px = 58 steps a pass at kU = 8 on the
// 4K bench frame; a region of 141 000 px was 276 steps a pass, 1.25 ms of a 13 ms frame -- components beyond 2^14 pixels go
// through fit_huge_kernel (same order of the additions, same bits).
template <int kU>
__device__ __forceinline__ void fit_large(const uint32_t* __restrict__ px, uint32_t comp, uint32_t off, uint32_t n, float s, float c,
                                          uint32_t uw, const float* __restrict__ dx, const float* __restrict__ dy,
                                          float* __restrict__ scratch_w, LineSegment* __restrict__ out, int lane) {
    float acc = 0.f;
    for (uint32_t i0 = lane; i0 < n; i0 += 64u * kU) {
        uint32_t p[kU];
        float gx[kU], gy[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            p[u] = i < n ? px[off + i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            gx[u] = dx[p[u]];
            gy[u] = dy[p[u]];
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            if (i < n) {
                const float wv = directional(gx[u], gy[u], s, c);
                scratch_w[off + i] = wv;
                acc = acc + wv;
            }
        }
    }
    const float S = wave_tree(acc);

    float ar = 0.f, ac = 0.f;
    for (uint32_t i0 = lane; i0 < n; i0 += 64u * kU) {
        uint32_t p[kU];
        float wq[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            p[u] = i < n ? px[off + i] : 0u;
            wq[u] = i < n ? scratch_w[off + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            if (i < n) {
                const float r = (float)(p[u] / uw), cc = (float)(p[u] % uw);
                const float wn = wq[u] / S;
                ar = ar + wn * r;
                ac = ac + wn * cc;
            }
        }
    }
    const float a_r = wave_tree(ar), a_c = wave_tree(ac);

    float crr = 0.f, crc = 0.f, ccc = 0.f;
    for (uint32_t i0 = lane; i0 < n; i0 += 64u * kU) {
        uint32_t p[kU];
        float wq[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            p[u] = i < n ? px[off + i] : 0u;
            wq[u] = i < n ? scratch_w[off + i] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            if (i < n) {
                const float cr = (float)(p[u] / uw) - a_r, cc = (float)(p[u] % uw) - a_c;
                const float wn = wq[u] / S;
                const float t = cr * wn, u2 = cc * wn;
                crr = crr + t * cr;
                crc = crc + t * cc;
                ccc = ccc + u2 * cc;
            }
        }
    }
    const float cov_rr = wave_tree(crr), cov_rc = wave_tree(crc), cov_cc = wave_tree(ccc);

    float d_r, d_c;
    major_axis_2x2(cov_rr, cov_rc, cov_cc, d_r, d_c);
    const float n_r = -d_c, n_c = d_r;

    float t0 = INFINITY, t1 = -INFINITY, es = 0.f;
    for (uint32_t i0 = lane; i0 < n; i0 += 64u * kU) {
        uint32_t p[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            p[u] = i < n ? px[off + i] : 0u;
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
            const uint32_t i = i0 + 64u * (uint32_t)u;
            if (i < n) {
                const float cr = (float)(p[u] / uw) - a_r, cc = (float)(p[u] % uw) - a_c;
                const float t = cr * d_r + cc * d_c;
                t0 = fminf(t0, t);
                t1 = fmaxf(t1, t);
                es = es + fabsf(cr * n_r + cc * n_c);
            }
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        t0 = fminf(t0, __shfl_xor(t0, o));
        t1 = fmaxf(t1, __shfl_xor(t1, o));
    }
    const float esum = wave_tree(es);
    if (lane == 0) {
        LineSegment l;
        l.x1 = a_c + d_c * t0;
        l.y1 = a_r + d_r * t0;
        l.x2 = a_c + d_c * t1;
        l.y2 = a_r + d_r * t1;
        l.weight = S / (float)n;
        l.err = esum / (float)n;
        l.group_id = -1;
        out[comp] = l;
    }
}

__global__ __launch_bounds__(256) void fit_kernel(const uint32_t* __restrict__ px, const uint32_t* __restrict__ px_unsorted,
                                                  const uint32_t* __restrict__ comp_off,
                                                  const uint32_t* __restrict__ comp_seed, const uint32_t* __restrict__ n_comp_ptr,
                                                  const int32_t* __restrict__ seed_bin, const float* __restrict__ dx,
                                                  const float* __restrict__ dy, int w, BinTrig trig,
                                                  float* __restrict__ scratch_w, LineSegment* __restrict__ out,
                                                  const uint32_t* __restrict__ huge_cursor) {
    const int lane = threadIdx.x & 63;
    const uint32_t comp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (comp >= *n_comp_ptr) return;
    const uint32_t off = comp_off[comp];
    const uint32_t n = comp_off[comp + 1] - off;
    const int b = seed_bin[comp_seed[comp]];
    const float s = trig.st[b], c = trig.ct[b];
    const uint32_t uw = (uint32_t)w;

    if (n <= 64u) {
        // A component of up to 64 pixels is one pixel per lane: everything after the first (dependent) loads stays in
        // registers -- the general form below reads the pixel list and the weights back from memory in each of its four
        // passes, a round trip each.  Same operations in the same order (the lane-strided sums have one term each: 0 + x),
        // so the same bits.  Most components are this small (4K bench frame: 19 000 of 20 500; natural frame: 46 000 of 47 400).
        const bool has = (uint32_t)lane < n;
        // the list in ascending pixel order (the order of the reference's component lists, which the sums follow): lane
        // `rank` gets this lane's key
        const uint32_t key = has ? px_unsorted[off + lane] : 0xFFFFFFFFu;
        uint32_t rank = 0;
#pragma unroll
        for (int j = 0; j < 64; ++j) rank += ((uint32_t)__builtin_amdgcn_readlane((int)key, j) < key) ? 1u : 0u;
        // (the padding keys are equal and all rank n: they collide on lane n or beyond, which no sum reads)
        const uint32_t p_sorted = (uint32_t)__builtin_amdgcn_ds_permute((int)(min(rank, 63u) << 2), (int)key);
        const uint32_t p = has ? p_sorted : 0u;
        float wv = 0.f;
        if (has) wv = directional(dx[p], dy[p], s, c);
        float acc = 0.f;
        if (has) acc = acc + wv;
        const float S = wave_tree(acc);
        const float r = (float)(p / uw), cq = (float)(p % uw);
        const float wn = wv / S;
        float ar = 0.f, ac = 0.f;
        if (has) {
            ar = ar + wn * r;
            ac = ac + wn * cq;
        }
        const float a_r = wave_tree(ar), a_c = wave_tree(ac);
        const float cr = r - a_r, cc = cq - a_c;
        float crr = 0.f, crc = 0.f, ccc = 0.f;
        if (has) {
            const float t = cr * wn, u = cc * wn;
            crr = crr + t * cr;
            crc = crc + t * cc;
            ccc = ccc + u * cc;
        }
        const float cov_rr = wave_tree(crr), cov_rc = wave_tree(crc), cov_cc = wave_tree(ccc);
        float d_r, d_c;
        major_axis_2x2(cov_rr, cov_rc, cov_cc, d_r, d_c);
        const float n_r = -d_c, n_c = d_r;
        float t0 = INFINITY, t1 = -INFINITY, es = 0.f;
        if (has) {
            const float t = cr * d_r + cc * d_c;
            t0 = fminf(t0, t);
            t1 = fmaxf(t1, t);
            es = es + fabsf(cr * n_r + cc * n_c);
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            t0 = fminf(t0, __shfl_xor(t0, o));
            t1 = fmaxf(t1, __shfl_xor(t1, o));
        }
        const float esum = wave_tree(es);
        if (lane == 0) {
            LineSegment l;
            l.x1 = a_c + d_c * t0;
            l.y1 = a_r + d_r * t0;
            l.x2 = a_c + d_c * t1;
            l.y2 = a_r + d_r * t1;
            l.weight = S / (float)n;
            l.err = esum / (float)n;
            l.group_id = -1;
            out[comp] = l;
        }
        return;
    }

    // (a component on the list of the huge ones -- its cursor word is its flagged number there -- is fit_huge_kernel's)
    if (huge_cursor != nullptr && n > kSortLdsBig && (huge_cursor[comp] & kHugeFlag) != 0u) return;
    fit_large<8>(px, comp, off, n, s, c, uw, dx, dy, scratch_w, out, lane);
}

// The components of more than 2^14 pixels (the list the offsets pass made of them), a workgroup of 1024 threads each.  The sums'
// canonical order is a single wavefront's -- lane L adds the elements L, L + 64, ... in that order, then the tree -- and one
// wavefront alone issues an instruction every four cycles: a region of 141 000 pixels was 0.93 ms of divisions, products and
// gathers for ONE wavefront (1.25 ms with eight pixels a lane in flight; thirty-two or sixty-four: no difference, it is issue,
// not latency).  So the work is split the other way: for a chunk of 4096 elements ALL sixteen wavefronts compute the elements'
// terms (the same expressions, so the same bits) into LDS, then wavefront w adds the chunk's terms of sum w in the canonical
// order -- lane L takes L, L + 64, ... of the chunk, chunks in ascending order: the same sequence of additions.
// lane L of a wavefront adds the chunk's terms L, L + 64, ... to its sum IN THAT ORDER; sixteen LDS reads are in flight at a time
// (one read and one dependent add a turn was 110 cycles a term: 2.9 of the 3.7 us a chunk took)
__device__ __forceinline__ float huge_chunk_sum(float acc, const float* __restrict__ v, uint32_t cn, int lane) {
    for (uint32_t j0 = (uint32_t)lane; j0 < cn; j0 += 64u * 16u) {
        float t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const uint32_t j = j0 + 64u * (uint32_t)u;
            t[u] = j < cn ? v[j] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (j0 + 64u * (uint32_t)u < cn) acc = acc + t[u];
    }
    return acc;
}
// LDS of the launch: 96 KB of terms (dynamic), cut into as many arrays as the pass has sums -- chunks of 24 576 elements for
// the passes with one sum, 12 288 for the centroid's two, 8 192 for the covariance's three: a chunk costs two barriers and a
// round trip to memory whatever its size (chunks of 4 096 for every pass: 3.7 us each, 140 of them for 141 000 pixels).
constexpr int kHugeTerms = 24576;
constexpr size_t kHugeLdsBytes = (size_t)kHugeTerms * sizeof(float);
__global__ __launch_bounds__(1024) void fit_huge_kernel(const uint32_t* __restrict__ px, const uint32_t* __restrict__ comp_off,
                                                        const uint32_t* __restrict__ comp_seed, const uint32_t* __restrict__ huge_list,
                                                        const uint32_t* __restrict__ n_large, uint32_t huge_max,
                                                        const int32_t* __restrict__ seed_bin, const float* __restrict__ dx,
                                                        const float* __restrict__ dy, int w, BinTrig trig,
                                                        float* __restrict__ scratch_w, LineSegment* __restrict__ out) {
    extern __shared__ float s_terms[];
    __shared__ float s_res[3];
    __shared__ float s_mm[2][16];
    if (blockIdx.x >= min(n_large[2], huge_max)) return;
    const uint32_t comp = huge_list[blockIdx.x];
    const uint32_t off = comp_off[comp];
    const uint32_t n = comp_off[comp + 1] - off;
    const int b = seed_bin[comp_seed[comp]];
    const float s = trig.st[b], c = trig.ct[b];
    const uint32_t uw = (uint32_t)w;
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = tid >> 6;

    // pass 1: weights (kept in scratch_w for the passes below) and their sum
    float acc = 0.f;
    {
        constexpr uint32_t kC = 16384u;  // (24 576 here: three arrays of 24 registers, 40 bytes a lane spilled)
        constexpr int kE = (int)(kC / 1024u);
        for (uint32_t c0 = 0; c0 < n; c0 += kC) {
            const uint32_t cn = min(kC, n - c0);
            uint32_t p[kE];
            float gx[kE], gy[kE];
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                p[e] = j < cn ? px[off + c0 + j] : 0u;
            }
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                gx[e] = dx[p[e]];
                gy[e] = dy[p[e]];
            }
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                if (j < cn) {
                    const float wv = directional(gx[e], gy[e], s, c);
                    scratch_w[off + c0 + j] = wv;
                    s_terms[j] = wv;
                }
            }
            __syncthreads();
            if (wave == 0) acc = huge_chunk_sum(acc, s_terms, cn, lane);
            __syncthreads();
        }
    }
    if (wave == 0) {
        const float t = wave_tree(acc);
        if (lane == 0) s_res[0] = t;
    }
    __syncthreads();
    const float S = s_res[0];
    __syncthreads();

    // pass 2: the weighted centroid (two sums: wavefronts 0 and 1)
    acc = 0.f;
    {
        constexpr uint32_t kC = kHugeTerms / 2;
        constexpr int kE = (int)(kC / 1024u);
        for (uint32_t c0 = 0; c0 < n; c0 += kC) {
            const uint32_t cn = min(kC, n - c0);
            uint32_t q[kE];
            float wq[kE];
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                q[e] = j < cn ? px[off + c0 + j] : 0u;
                wq[e] = j < cn ? scratch_w[off + c0 + j] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                if (j < cn) {
                    const float r = (float)(q[e] / uw), cc = (float)(q[e] % uw);
                    const float wn = wq[e] / S;
                    s_terms[j] = wn * r;
                    s_terms[kC + j] = wn * cc;
                }
            }
            __syncthreads();
            if (wave < 2) acc = huge_chunk_sum(acc, s_terms + (size_t)wave * kC, cn, lane);
            __syncthreads();
        }
    }
    if (wave < 2) {
        const float t = wave_tree(acc);
        if (lane == 0) s_res[wave] = t;
    }
    __syncthreads();
    const float a_r = s_res[0], a_c = s_res[1];
    __syncthreads();

    // pass 3: the covariance (three sums: wavefronts 0, 1, 2)
    acc = 0.f;
    {
        constexpr uint32_t kC = kHugeTerms / 3;
        constexpr int kE = (int)(kC / 1024u);
        static_assert(kC % 1024u == 0u, "a chunk is a whole number of elements per thread");
        for (uint32_t c0 = 0; c0 < n; c0 += kC) {
            const uint32_t cn = min(kC, n - c0);
            uint32_t q[kE];
            float wq[kE];
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                q[e] = j < cn ? px[off + c0 + j] : 0u;
                wq[e] = j < cn ? scratch_w[off + c0 + j] : 0.f;
            }
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                if (j < cn) {
                    const float cr = (float)(q[e] / uw) - a_r, cc = (float)(q[e] % uw) - a_c;
                    const float wn = wq[e] / S;
                    const float t = cr * wn, u2 = cc * wn;
                    s_terms[j] = t * cr;
                    s_terms[kC + j] = t * cc;
                    s_terms[2u * kC + j] = u2 * cc;
                }
            }
            __syncthreads();
            if (wave < 3) acc = huge_chunk_sum(acc, s_terms + (size_t)wave * kC, cn, lane);
            __syncthreads();
        }
    }
    if (wave < 3) {
        const float t = wave_tree(acc);
        if (lane == 0) s_res[wave] = t;
    }
    __syncthreads();
    const float cov_rr = s_res[0], cov_rc = s_res[1], cov_cc = s_res[2];
    __syncthreads();

    float d_r, d_c;
    major_axis_2x2(cov_rr, cov_rc, cov_cc, d_r, d_c);
    const float n_r = -d_c, n_c = d_r;

    // pass 4: the extent along the axis (minimum and maximum: any order) and the summed distance from it (wavefront 0)
    float t0 = INFINITY, t1 = -INFINITY;
    acc = 0.f;
    {
        constexpr uint32_t kC = kHugeTerms;
        constexpr int kE = (int)(kC / 1024u);
        for (uint32_t c0 = 0; c0 < n; c0 += kC) {
            const uint32_t cn = min(kC, n - c0);
            uint32_t q[kE];
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                q[e] = j < cn ? px[off + c0 + j] : 0u;
            }
#pragma unroll
            for (int e = 0; e < kE; ++e) {
                const uint32_t j = (uint32_t)tid + 1024u * (uint32_t)e;
                if (j < cn) {
                    const float cr = (float)(q[e] / uw) - a_r, cc = (float)(q[e] % uw) - a_c;
                    const float t = cr * d_r + cc * d_c;
                    t0 = fminf(t0, t);
                    t1 = fmaxf(t1, t);
                    s_terms[j] = fabsf(cr * n_r + cc * n_c);
                }
            }
            __syncthreads();
            if (wave == 0) acc = huge_chunk_sum(acc, s_terms, cn, lane);
            __syncthreads();
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        t0 = fminf(t0, __shfl_xor(t0, o));
        t1 = fmaxf(t1, __shfl_xor(t1, o));
    }
    if (lane == 0) {
        s_mm[0][wave] = t0;
        s_mm[1][wave] = t1;
    }
    __syncthreads();
    if (wave != 0) return;
    const float esum = wave_tree(acc);
    if (lane == 0) {
        for (int i = 1; i < 16; ++i) {
            t0 = fminf(t0, s_mm[0][i]);
            t1 = fmaxf(t1, s_mm[1][i]);
        }
        LineSegment l;
        l.x1 = a_c + d_c * t0;
        l.y1 = a_r + d_r * t0;
        l.x2 = a_c + d_c * t1;
        l.y2 = a_r + d_r * t1;
        l.weight = S / (float)n;
        l.err = esum / (float)n;
        l.group_id = -1;
        out[comp] = l;
    }
}

}  // namespace

size_t fit_temp_bytes(size_t max_pixels, uint32_t max_segments) {
    (void)max_segments;
    // status words of the component scan (16 B per 2048 seeds; zero when allocated: component_offsets_kernel); seeds <= pixels
    return (max_pixels / kOffChunk + 2) * 2 * sizeof(unsigned long long) + 256;
}

int launch_component_offsets(const int32_t* seed_size, const uint32_t* d_n_seeds, uint32_t seed_cap, int min_size,
                             uint32_t* comp_rank, uint32_t* comp_seed, uint32_t* comp_off, uint32_t* totals,
                             uint32_t* large_list, uint32_t large_cap, uint32_t* n_large, void* temp, size_t temp_bytes,
                             uint32_t frame_tag, uint32_t* cursor, const HugeSort& hs, hipStream_t s) {
    const uint32_t chunks = (seed_cap + kOffChunk - 1) / kOffChunk;
    if (chunks == 0 || temp_bytes < (size_t)chunks * 2 * sizeof(unsigned long long)) {
        set_error("launch_component_offsets: workspace too small");
        return 1;
    }
    hipLaunchKernelGGL(component_offsets_kernel, dim3(chunks), dim3(256), 0, s, seed_size, d_n_seeds, seed_cap, min_size,
                       static_cast<unsigned long long*>(temp), frame_tag, comp_rank, comp_seed, comp_off, totals, large_list, large_cap,
                       n_large, cursor, hs.list, hs.tab ? hs.max : 0u);
    LR_HIP(hipGetLastError());
    return 0;
}

// `with_huge`: false when the caller KNOWS that the frame has no flood of more than 2^14 pixels (the flood's rounds report
// their largest commit to the host that enqueues them just in time): the counting launch is left out.
int launch_component_scatter(const uint32_t* label, size_t npix, const uint32_t* comp_rank, const uint32_t* comp_off,
                             uint32_t* cursor, uint32_t* px, const HugeSort& hs, uint32_t* n_large, bool with_huge, hipStream_t s) {
    const uint32_t nb = (uint32_t)((npix + ((size_t)1 << kHugeShift) - 1) >> kHugeShift);
    const size_t want = (npix + 255) / 256;
    if (with_huge && hs.tab)
        hipLaunchKernelGGL(huge_count_kernel, dim3((unsigned)std::min<size_t>(want, 2048)), dim3(256), 0, s, label, npix, comp_rank, comp_off,
                           cursor, hs.tab, nb, hs.list, hs.max, reinterpret_cast<uint4*>(hs.jobs), n_large);
    const int blocks = (int)(want < 8192 ? want : 8192);
    hipLaunchKernelGGL(component_scatter_kernel, dim3(blocks > 0 ? blocks : 1), dim3(256), 0, s, label, npix, comp_rank,
                       comp_off, cursor, px, hs.tab, nb);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_component_sort(const uint32_t* px_in, uint32_t* px_out, const uint32_t* comp_off, const uint32_t* d_n_comp,
                          uint32_t comp_cap, const uint32_t* large_list, uint32_t large_cap, const uint32_t* n_large,
                          uint32_t* scratch, const uint32_t* cursor, const HugeSort& hs, bool with_huge, hipStream_t s) {
    if (comp_cap == 0) return 0;
    (void)d_n_comp;  // (lists of up to 64 pixels: fit_kernel)
    hipLaunchKernelGGL(component_sort_large_kernel, dim3(4096), dim3(256), 0, s, px_in, px_out, comp_off, large_list, n_large);
    hipLaunchKernelGGL(component_sort_big_kernel<false>, dim3(128), dim3(1024), 0, s, px_in, px_out, comp_off, large_list,
                       large_cap, n_large, scratch, cursor);
    if (with_huge && hs.tab) {
        hipLaunchKernelGGL(huge_sort_kernel, dim3(512), dim3(1024), 0, s, px_in, px_out, reinterpret_cast<const uint4*>(hs.jobs), hs.tab, n_large);
        // (more huge components than the table has rows for -- never: the old way, one workgroup through global memory)
        hipLaunchKernelGGL(component_sort_big_kernel<true>, dim3(1), dim3(1024), 0, s, px_in, px_out, comp_off, large_list,
                           large_cap, n_large, scratch, cursor);
    }
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_fit(const uint32_t* px_sorted, const uint32_t* px_unsorted, const uint32_t* comp_off, const uint32_t* comp_seed, const uint32_t* d_n_comp,
               uint32_t comp_cap, const int32_t* seed_bin, const float* dx, const float* dy, int w, BinTrig trig,
               float* scratch_w, LineSegment* out, const uint32_t* cursor, const HugeSort& hs, const uint32_t* n_large, bool with_huge,
               hipStream_t s) {
    if (comp_cap == 0) return 0;
    const bool huge = with_huge && hs.tab != nullptr && hs.max != 0u;
    if (huge) {  // (first: the launch lasts as long as its largest component, the small ones' launch runs behind it)
        // the opt-in for more than 64 KB of dynamic LDS is a per-device attribute of the kernel: once per device of this process
        static std::atomic<uint64_t> done_mask{0};
        int dev = 0;
        LR_HIP(hipGetDevice(&dev));
        const uint64_t bit = 1ull << (dev & 63);
        if (!(done_mask.load(std::memory_order_acquire) & bit)) {
            LR_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(fit_huge_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kHugeLdsBytes));
            done_mask.fetch_or(bit, std::memory_order_release);
        }
        hipLaunchKernelGGL(fit_huge_kernel, dim3(hs.max), dim3(1024), kHugeLdsBytes, s, px_sorted, comp_off, comp_seed, hs.list, n_large, hs.max, seed_bin,
                           dx, dy, w, trig, scratch_w, out);
    }
    hipLaunchKernelGGL(fit_kernel, dim3((comp_cap + 3) / 4), dim3(256), 0, s, px_sorted, px_unsorted, comp_off, comp_seed, d_n_comp,
                       seed_bin, dx, dy, w, trig, scratch_w, out, huge ? cursor : nullptr);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
