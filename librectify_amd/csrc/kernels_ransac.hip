// RANSAC hypothesis scoring for the line-pencil (vanishing point) model.
//
// Reference: RANSAC_Estimator::solve (estimator.h:37-71) draws a 2-line sample, rejects it if
// the two normalised homogeneous lines are closer than 0.05 (line_pencil.cpp:89-98), takes
// their cross product as the hypothesis (:101-108) and scores it with the length-weighted
// inlier sum over the remaining lines (:131-140, geometry.cpp:214-229); the first strictly
// best hypothesis wins (estimator.h:62-70).
//
// Here one wavefront owns one hypothesis: the sample is derived from (seed, round, iteration)
// by a counter-based generator, so no sequential RNG state exists; the 64 lanes stride over
// the lines and the score is reduced with the canonical tree T().  The line table
// (anchor, direction, length: 20 B/line) is read through L1/L2 by every wavefront; there is
// no HBM traffic to speak of and no dense contraction, hence no MFMA.
#include "common.h"

namespace lramd {
namespace {

__device__ inline float wave_tree(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

__global__ __launch_bounds__(256) void ransac_score_kernel(PencilSoA m, uint32_t n, float tol, float degeneracy_tol,
                                                           uint32_t n_iter, uint64_t seed, uint32_t round,
                                                           float* __restrict__ scores) {
    const int lane = threadIdx.x & 63;
    const uint32_t hyp = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (hyp >= n_iter) return;
    uint32_t a, b;
    sample_pair(seed, round, hyp, n, a, b);
    const float hax = m.hx[a], hay = m.hy[a], haz = m.hz[a];
    const float hbx = m.hx[b], hby = m.hy[b], hbz = m.hz[b];
    const float ex = hax - hbx, ey = hay - hby, ez = haz - hbz;
    const float dist = sqrtf((ex * ex + ey * ey) + ez * ez);
    if (!(dist > degeneracy_tol)) {  // sample_check
        if (lane == 0) scores[hyp] = -1.0f;
        return;
    }
    // fit: h_a x h_b
    const float px = hay * hbz - haz * hby;
    const float py = haz * hbx - hax * hbz;
    const float pz = hax * hby - hay * hbx;
    const bool ideal = fabsf(pz) < kEps;  // inclination(): ideal point is used as a direction
    const float pnx = px / pz, pny = py / pz;
    float acc = 0.f;
    for (uint32_t i = lane; i < n; i += 64) {
        const float vx = ideal ? px : (pnx - m.ax[i]);
        const float vy = ideal ? py : (pny - m.ay[i]);
        const float nn = vx * vx + vy * vy;
        const float nrm = sqrtf(nn);
        const float ux = vx / nrm, uy = vy / nrm;
        const float inc = fabsf(ux * m.dx[i] + uy * m.dy[i]);
        const float err = -inc + 1.0f;
        acc = acc + ((err < tol) ? m.len[i] : 0.0f);
    }
    const float score = wave_tree(acc);
    if (lane == 0) scores[hyp] = score;
}

// First strictly best hypothesis: max score, ties to the lowest iteration; -1 if none scored > 0.
__global__ __launch_bounds__(1024) void ransac_argmax_kernel(const float* __restrict__ scores, uint32_t n_iter,
                                                             float* __restrict__ best_score,
                                                             int32_t* __restrict__ best_iter) {
    __shared__ float s_v[16];
    __shared__ int s_i[16];
    float bv = 0.f;
    int bi = -1;
    for (uint32_t i = threadIdx.x; i < n_iter; i += 1024) {
        const float v = scores[i];
        if (v > bv) {
            bv = v;
            bi = (int)i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(bv, off);
        const int oi = __shfl_xor(bi, off);
        if (ov > bv || (ov == bv && oi >= 0 && (bi < 0 || oi < bi))) {
            bv = ov;
            bi = oi;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_v[threadIdx.x >> 6] = bv;
        s_i[threadIdx.x >> 6] = bi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) {
            const float ov = s_v[k];
            const int oi = s_i[k];
            if (ov > bv || (ov == bv && oi >= 0 && (bi < 0 || oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        *best_score = bv;
        *best_iter = bi;
    }
}

}  // namespace

int launch_ransac_score(PencilSoA m, uint32_t n, float tol, float degeneracy_tol, uint32_t n_iter, uint64_t seed,
                        uint32_t round, float* scores, hipStream_t s) {
    if (n < 2 || n_iter == 0) {
        set_error("launch_ransac_score: need at least 2 lines and 1 iteration");
        return 1;
    }
    hipLaunchKernelGGL(ransac_score_kernel, dim3((n_iter + 3) / 4), dim3(256), 0, s, m, n, tol, degeneracy_tol, n_iter,
                       seed, round, scores);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_ransac_argmax(const float* scores, uint32_t n_iter, float* best_score, int32_t* best_iter, hipStream_t s) {
    hipLaunchKernelGGL(ransac_argmax_kernel, dim3(1), dim3(1024), 0, s, scores, n_iter, best_score, best_iter);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
