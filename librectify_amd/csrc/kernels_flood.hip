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
                                                           const float* __restrict__ seed_thr, uint32_t n_seeds,
                                                           BinTrig trig, uint32_t* label, int32_t* seed_size,
                                                           int32_t* queue) {
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
                         const int32_t* seed_bin, const float* seed_thr, uint32_t n_seeds, BinTrig trig,
                         uint32_t* label, int32_t* seed_size, int32_t* queue, hipStream_t s) {
    (void)h;
    if (n_seeds == 0) return 0;
    hipLaunchKernelGGL(flood_ordered_kernel, dim3(1), dim3(64), 0, s, dx, dy, dmask, w, seed_idx, seed_bin, seed_thr,
                       n_seeds, trig, label, seed_size, queue);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
