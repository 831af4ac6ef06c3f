// Line filtering and vanishing-point grouping on the device, so that a frame needs no host round trip between the
// line fit and its result.
//
// Reference: _filter_lines (interface.cpp:26-32), the "fewer than two raw segments" exit (interface.cpp:50-54),
// estimate_line_pencils (line_pencil.cpp:148-177): bounding-box normalisation (geometry.cpp:96-112,258-282), the
// LinePencilModel constructor (line_pencil.cpp:25-32), and estimate_multiple_structures (estimator.h:99-145) around
// RANSAC_Estimator::solve (estimator.h:37-78): at most four peeling rounds of
//     score 10 000 two-line hypotheses (kernels_ransac.hip) -> first strictly best -> inliers of it -> fit_optimal
//     (line_pencil.cpp:111-128: 3x3 length-weighted scatter of the inliers' homogeneous lines, eigenvector of the
//     smallest eigenvalue) -> inliers of the refit get the round's id, near misses become garbage, the rest goes on.
// The rounds are sequential by nature; each is one scoring launch plus one single-workgroup "peel" kernel, and a
// round that has nothing left to do (fewer than two lines remain) leaves at once.  All arithmetic is the canonical
// fp32 form of DESIGN.md §3 (sums over lines are the tree T(); the 3x3 eigenproblem is cyclic Jacobi in double).
#include "common.h"

namespace lramd {
namespace {

constexpr int kWG = 1024;
constexpr int kWaves = kWG / 64;

__device__ __forceinline__ float wave_tree(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

// position of this thread's element among the flagged elements of the workgroup's current chunk (in thread order),
// and the chunk's total.  s_cnt: kWaves words of LDS.
__device__ __forceinline__ uint32_t block_rank(bool flag, uint32_t* s_cnt, uint32_t& total) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint64_t m = __ballot(flag);
    __syncthreads();  // s_cnt of the previous call has been read by everyone
    if (lane == 0) s_cnt[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    uint32_t before = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < kWaves; ++i) {
        const uint32_t c = s_cnt[i];
        before += i < wv ? c : 0u;
        tot += c;
    }
    total = tot;
    return before + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
}

// pencil model of one segment (line_pencil.cpp:25-32 on the normalised segments): into the table of ALL lines (kept for
// fit_optimal's "empty set = every line" case) and the identical compacted table of round 0
__device__ __forceinline__ void pencil_model_of(const LineSegment& l, uint32_t i, float cx, float cy, float sc, const PencilTable& all,
                                                const PencilTable& round0) {
    const float x1 = (l.x1 - cx) / sc, y1 = (l.y1 - cy) / sc, x2 = (l.x2 - cx) / sc, y2 = (l.y2 - cy) / sc;
    // h = unit((x1, y1, 1) x (x2, y2, 1)) (geometry.cpp:64-69); Eigen's normalized() leaves a zero vector alone
    float hx = y1 * 1.f - 1.f * y2, hy = 1.f * x2 - x1 * 1.f, hz = x1 * y2 - y1 * x2;
    const float zz = (hx * hx + hy * hy) + hz * hz;
    if (zz > 0.0f) {
        const float nn = sqrtf(zz);
        hx = hx / nn;
        hy = hy / nn;
        hz = hz / nn;
    }
    const float ax = (x2 + x1) / 2, ay = (y2 + y1) / 2;  // geometry.cpp:72-75
    const float dx = x2 - x1, dy = y2 - y1;              // geometry.cpp:78-81
    const float len = sqrtf(dx * dx + dy * dy);
    const float ux = dx / len, uy = dy / len;
    all.ax[i] = ax; all.ay[i] = ay; all.dx[i] = ux; all.dy[i] = uy; all.len[i] = len;
    all.hx[i] = hx; all.hy[i] = hy; all.hz[i] = hz; all.orig[i] = i;
    round0.ax[i] = ax; round0.ay[i] = ay; round0.dx[i] = ux; round0.dy[i] = uy; round0.len[i] = len;
    round0.hx[i] = hx; round0.hy[i] = hy; round0.hz[i] = hz; round0.orig[i] = i;
}

// ---- filter_lines + bounding box ----------------------------------------------------------------------------
// One workgroup: stable compaction of the fitted segments that are long and straight enough, then the bounding box of
// what is left (min / max are exact whatever the order).  gctl: see GroupCtl in common.h.
__global__ __launch_bounds__(kWG) void filter_lines_kernel(const LineSegment* __restrict__ raw,
                                                           const uint32_t* __restrict__ n_raw_ptr, uint32_t raw_cap,
                                                           float min_length, LineSegment* __restrict__ out,
                                                           uint32_t* __restrict__ gctl, float* __restrict__ gnorm,
                                                           PencilTable all, PencilTable round0, uint32_t with_model) {
    // (kFlB chunks of kWG segments at a time: their loads are in flight together and the chunks' counts are exchanged behind
    // ONE barrier -- a chunk at a time was a dependent load and two barriers per 1024 segments, 22 us for the 20 000 of a 4K
    // frame in a launch of one workgroup)
    constexpr int kFlB = 8;
    __shared__ uint32_t s_cnt[kFlB][kWaves];
    __shared__ float s_red[4][kWaves];
    const uint32_t n_raw = min(*n_raw_ptr, raw_cap);
    const float ml = fmaxf(min_length, kLineMinLength);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    uint32_t base = 0;
    if (n_raw >= 2u) {  // interface.cpp:50-54: fewer than two raw segments -> nothing
        for (uint32_t i0 = 0; i0 < n_raw; i0 += kWG * kFlB) {
            LineSegment l[kFlB];
            uint64_t m[kFlB];
#pragma unroll
            for (int b = 0; b < kFlB; ++b) {
                const uint32_t i = i0 + (uint32_t)b * kWG + threadIdx.x;
                l[b] = LineSegment{};
                if (i < n_raw) l[b] = raw[i];
            }
#pragma unroll
            for (int b = 0; b < kFlB; ++b) {
                const uint32_t i = i0 + (uint32_t)b * kWG + threadIdx.x;
                const float dx = l[b].x2 - l[b].x1, dy = l[b].y2 - l[b].y1;
                const bool keep = i < n_raw && sqrtf(dx * dx + dy * dy) > ml && l[b].err < kLineMaxErr;
                m[b] = __ballot(keep);
                if (lane == 0) s_cnt[b][wv] = (uint32_t)__popcll(m[b]);
            }
            __syncthreads();
#pragma unroll
            for (int b = 0; b < kFlB; ++b) {
                uint32_t before = 0, tot = 0;
#pragma unroll
                for (int w = 0; w < kWaves; ++w) {
                    const uint32_t c = s_cnt[b][w];
                    before += w < wv ? c : 0u;
                    tot += c;
                }
                if ((m[b] >> lane) & 1ull) {
                    l[b].group_id = -1;
                    out[base + before + (uint32_t)__popcll(m[b] & ((1ull << lane) - 1ull))] = l[b];
                }
                base += tot;
            }
            __syncthreads();  // (the counts are rewritten by the next pass)
        }
    }
    __syncthreads();  // the workgroup's own stores to `out` are visible to it from here on
    const uint32_t n = base;
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (uint32_t i = threadIdx.x; i < n; i += kWG) {
        const LineSegment l = out[i];
        mnx = fminf(mnx, fminf(l.x1, l.x2));
        mny = fminf(mny, fminf(l.y1, l.y2));
        mxx = fmaxf(mxx, fmaxf(l.x1, l.x2));
        mxy = fmaxf(mxy, fmaxf(l.y1, l.y2));
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, off));
        mny = fminf(mny, __shfl_xor(mny, off));
        mxx = fmaxf(mxx, __shfl_xor(mxx, off));
        mxy = fmaxf(mxy, __shfl_xor(mxy, off));
    }
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = mnx;
        s_red[1][threadIdx.x >> 6] = mny;
        s_red[2][threadIdx.x >> 6] = mxx;
        s_red[3][threadIdx.x >> 6] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kWaves; ++i) {
            mnx = fminf(mnx, s_red[0][i]);
            mny = fminf(mny, s_red[1][i]);
            mxx = fmaxf(mxx, s_red[2][i]);
            mxy = fmaxf(mxy, s_red[3][i]);
        }
        const float sx = mxx - mnx, sy = mxy - mny;  // geometry.cpp:96-112,272-282
        gnorm[0] = mnx + 0.5f * sx;
        gnorm[1] = mny + 0.5f * sy;
        gnorm[2] = fmaxf(sx, sy);
        gctl[kGcLines] = n;
        gctl[kGcRemaining] = n;
        gctl[kGcRound] = 0u;
        gctl[kGcActive] = n;  // lines in the compacted table of the coming round
        s_red[0][0] = gnorm[0];
        s_red[1][0] = gnorm[1];
        s_red[2][0] = gnorm[2];
    }
    // Round 5: the pencil model of the kept segments in the same launch (it was one of its own, 5 us behind this one; a frame
    // keeps one or two thousand segments: a pass or two of the workgroup).
    if (with_model == 0u) return;
    __syncthreads();
    const float cx = s_red[0][0], cy = s_red[1][0], sc = s_red[2][0];
    for (uint32_t i = threadIdx.x; i < n; i += kWG) pencil_model_of(out[i], i, cx, cy, sc, all, round0);
}

// Same for lines that are already filtered (the refine path and lr_estimate_line_pencils upload them): bounding box
// and control words only.
__global__ __launch_bounds__(kWG) void lines_bbox_kernel(LineSegment* __restrict__ lines, uint32_t n,
                                                         uint32_t* __restrict__ gctl, float* __restrict__ gnorm) {
    __shared__ float s_red[4][kWaves];
    float mnx = INFINITY, mny = INFINITY, mxx = -INFINITY, mxy = -INFINITY;
    for (uint32_t i = threadIdx.x; i < n; i += kWG) {
        const LineSegment l = lines[i];
        mnx = fminf(mnx, fminf(l.x1, l.x2));
        mny = fminf(mny, fminf(l.y1, l.y2));
        mxx = fmaxf(mxx, fmaxf(l.x1, l.x2));
        mxy = fmaxf(mxy, fmaxf(l.y1, l.y2));
        lines[i].group_id = -1;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        mnx = fminf(mnx, __shfl_xor(mnx, off));
        mny = fminf(mny, __shfl_xor(mny, off));
        mxx = fmaxf(mxx, __shfl_xor(mxx, off));
        mxy = fmaxf(mxy, __shfl_xor(mxy, off));
    }
    if ((threadIdx.x & 63) == 0) {
        s_red[0][threadIdx.x >> 6] = mnx;
        s_red[1][threadIdx.x >> 6] = mny;
        s_red[2][threadIdx.x >> 6] = mxx;
        s_red[3][threadIdx.x >> 6] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < kWaves; ++i) {
            mnx = fminf(mnx, s_red[0][i]);
            mny = fminf(mny, s_red[1][i]);
            mxx = fmaxf(mxx, s_red[2][i]);
            mxy = fmaxf(mxy, s_red[3][i]);
        }
        const float sx = mxx - mnx, sy = mxy - mny;
        gnorm[0] = mnx + 0.5f * sx;
        gnorm[1] = mny + 0.5f * sy;
        gnorm[2] = fmaxf(sx, sy);
        gctl[kGcLines] = n;
        gctl[kGcRemaining] = n;
        gctl[kGcRound] = 0u;
        gctl[kGcActive] = n;
    }
}

// ---- pencil model (line_pencil.cpp:25-32 on the normalised segments) -----------------------------------------------
// Writes the table of ALL lines (kept for fit_optimal's "empty set = every line" case) and the identical compacted
// table of round 0.
__global__ __launch_bounds__(256) void pencil_model_kernel(const LineSegment* __restrict__ lines,
                                                           const uint32_t* __restrict__ gctl,
                                                           const float* __restrict__ gnorm, PencilTable all,
                                                           PencilTable round0) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= gctl[kGcLines]) return;
    pencil_model_of(lines[i], i, gnorm[0], gnorm[1], gnorm[2], all, round0);
}

// error of line (ax, ay, dx, dy) against hypothesis p (line_pencil.cpp:131-134, geometry.cpp:214-229)
__device__ __forceinline__ float pencil_error(float ax, float ay, float dx, float dy, float px, float py, float pz) {
    float vx, vy;
    if (fabsf(pz) < kEps) {
        vx = px;
        vy = py;
    } else {
        const float qx = px / pz, qy = py / pz;
        vx = qx - ax;
        vy = qy - ay;
    }
    const float nn = vx * vx + vy * vy;
    const float nrm = sqrtf(nn);
    const float ux = vx / nrm, uy = vy / nrm;
    return -fabsf(ux * dx + uy * dy) + 1.0f;
}

// Smallest-eigenvalue eigenvector of a symmetric 3x3 (float in, cyclic Jacobi in double): the same operations in the
// same order as the oracle's and the host's (vp_host.cpp: smallest_eigenvector).
__device__ void smallest_eigenvector_3x3(const float c[9], float out[3]) {
    double A[3][3];
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = 0.5 * ((double)c[i * 3 + j] + (double)c[j * 3 + i]);
    for (int sweep = 0; sweep < 64; ++sweep) {
        const double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        // converged: the off-diagonal part is below double precision relative to the diagonal (waiting for it to
        // underflow to exactly zero can take all 64 sweeps and changes nothing in the float result)
        const double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-36 * dg) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double cs = 1.0 / sqrt(t * t + 1.0), sn = t * cs;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = cs * akp - sn * akq;
                    A[k][q] = sn * akp + cs * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = cs * apk - sn * aqk;
                    A[q][k] = sn * apk + cs * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = cs * vkp - sn * vkq;
                    V[k][q] = sn * vkp + cs * vkq;
                }
            }
    }
    int k = 0;
    if (A[1][1] < A[k][k]) k = 1;
    if (A[2][2] < A[k][k]) k = 2;
    const double nx = V[0][k], ny = V[1][k], nz = V[2][k];
    const double nn = sqrt(nx * nx + ny * ny + nz * nz);
    out[0] = (float)(nx / nn);
    out[1] = (float)(ny / nn);
    out[2] = (float)(nz / nn);
}

// ---- one peeling round after its hypotheses have been scored (estimator.h:62-76,118-139) -----------------------------
// cur: the compacted table the scores refer to (kGcActive lines); nxt: where the lines that go on to the next round
// are written, in order; all: every line of the model; stage_g: scratch for the inliers' (h, length) beyond the first
// kStage, which live in LDS (>= 4 * kGcActive floats).
// The kernel is a chain of short dependent phases, so what counts is the number of memory round trips: a thread keeps
// its line of the first 1024 in registers from the start, and the inliers are staged in LDS for the refit's sums.
constexpr uint32_t kStage = 2048;
__device__ __forceinline__ void peel_body(const PencilTable& cur, const PencilTable& nxt, const PencilTable& all,
                                          unsigned long long* __restrict__ best_slots, uint64_t seed,
                                          float tol, float garbage_tol, int max_models,
                                          uint32_t* __restrict__ gctl, float4* __restrict__ stage_g,
                                          LineSegment* __restrict__ lines, float* __restrict__ models) {
    __shared__ uint32_t s_cnt[kWaves];
    __shared__ float s_bv[kWaves];
    __shared__ int s_bi[kWaves];
    __shared__ float s_h[3];
    __shared__ float4 s_stage[kStage];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t n = gctl[kGcActive];
    const uint32_t round = gctl[kGcRound];
    // estimator.h:115: while (num_observations >= minimum_set_size && k < max_structures)
    if (gctl[kGcRemaining] < 2u || round >= (uint32_t)max_models) return;  // (the round's scoring launch left as well: the slots are clear)

#ifdef LR_PEEL_TIMING
    unsigned long long tm[6];
    tm[0] = wall_clock64();
#endif
    // this thread's line of the first chunk, in flight while the best hypothesis is looked up
    float r_ax = 0.f, r_ay = 0.f, r_dx = 0.f, r_dy = 0.f, r_len = 0.f, r_hx = 0.f, r_hy = 0.f, r_hz = 0.f;
    uint32_t r_orig = 0;
    if (threadIdx.x < n) {
        const uint32_t i = threadIdx.x;
        r_ax = cur.ax[i]; r_ay = cur.ay[i]; r_dx = cur.dx[i]; r_dy = cur.dy[i]; r_len = cur.len[i];
        r_hx = cur.hx[i]; r_hy = cur.hy[i]; r_hz = cur.hz[i]; r_orig = cur.orig[i];
    }

    // -- first strictly best hypothesis: highest score, lowest iteration among equals; none if no score is positive.  The
    // scoring launch has reduced it into kRansacBestSlots words (score bits : 32 | ~iteration : 32, kernels_ransac.hip);
    // they are read and cleared here for the next round's scoring.  (Without a scoring launch they are zero: none.)
    if (threadIdx.x < 64) {
        unsigned long long key = threadIdx.x < (unsigned)kRansacBestSlots ? best_slots[threadIdx.x] : 0ull;
        if (threadIdx.x < (unsigned)kRansacBestSlots) best_slots[threadIdx.x] = 0ull;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), off) << 32) |
                                         (uint32_t)__shfl_xor((int)(uint32_t)key, off);
            key = o > key ? o : key;
        }
        if (threadIdx.x == 0) {
            s_bv[0] = __uint_as_float((uint32_t)(key >> 32));
            s_bi[0] = key ? (int)(0xFFFFFFFFu - (uint32_t)key) : -1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int bi = s_bi[0];
        // the hypothesis itself: h_a x h_b of the winning sample (line_pencil.cpp:101-108); (0, 0, 0) if there is none
        // (the reference leaves it uninitialised, estimator.h:39)
        float hx = 0.f, hy = 0.f, hz = 0.f;
        if (bi >= 0) {
            uint32_t a, b;
            sample_pair(seed, round, (uint32_t)bi, n, a, b);
            const float ax = cur.hx[a], ay = cur.hy[a], az = cur.hz[a];
            const float bx = cur.hx[b], by = cur.hy[b], bz = cur.hz[b];
            hx = ay * bz - az * by;
            hy = az * bx - ax * bz;
            hz = ax * by - ay * bx;
        }
        s_h[0] = hx;
        s_h[1] = hy;
        s_h[2] = hz;
        if (round < (uint32_t)(kGcWords - kGcBestIter0)) gctl[kGcBestIter0 + round] = (uint32_t)bi;  // (diagnostics: four slots)
    }
    __syncthreads();
    const float bx_ = s_h[0], by_ = s_h[1], bz_ = s_h[2];
#ifdef LR_PEEL_TIMING
    tm[1] = wall_clock64();
#endif

    // -- inliers of the best hypothesis, in order (estimator.h:74-75): their (h, length) staged for the refit
    uint32_t n_inl = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += kWG) {
        const uint32_t i = i0 + threadIdx.x;
        bool in = false;
        float4 hl = make_float4(r_hx, r_hy, r_hz, r_len);
        if (i < n) {
            if (i0 == 0) {
                in = pencil_error(r_ax, r_ay, r_dx, r_dy, bx_, by_, bz_) < tol;
            } else {
                in = pencil_error(cur.ax[i], cur.ay[i], cur.dx[i], cur.dy[i], bx_, by_, bz_) < tol;
                if (in) hl = make_float4(cur.hx[i], cur.hy[i], cur.hz[i], cur.len[i]);
            }
        }
        uint32_t tot;
        const uint32_t r = block_rank(in, s_cnt, tot);
        if (in) {
            const uint32_t o = n_inl + r;
            if (o < kStage) s_stage[o] = hl;
            else stage_g[o] = hl;
        }
        n_inl += tot;
    }
    __syncthreads();

#ifdef LR_PEEL_TIMING
    tm[2] = wall_clock64();
#endif
    // -- fit_optimal (line_pencil.cpp:111-128): cov = sum_i (h_i * len_i) h_i^T over the inliers, or over EVERY line of
    // the model if there is none (the reference's "empty index set means all"); nine tree sums by the first wavefront
    if (wv == 0) {
        const uint32_t m = n_inl ? n_inl : gctl[kGcLines];
        float acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.f;
        for (uint32_t j = lane; j < m; j += 64) {
            float4 hl;
            if (n_inl) hl = j < kStage ? s_stage[j] : stage_g[j];
            else hl = make_float4(all.hx[j], all.hy[j], all.hz[j], all.len[j]);
            const float hv[3] = {hl.x, hl.y, hl.z};
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const float t = hv[a] * hl.w;
#pragma unroll
                for (int b = 0; b < 3; ++b) acc[a * 3 + b] = acc[a * 3 + b] + t * hv[b];
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = wave_tree(acc[k]);
#ifdef LR_PEEL_TIMING
        tm[3] = wall_clock64();
#endif
        if (lane == 0) {
            float hf[3];
            smallest_eigenvector_3x3(acc, hf);
            s_h[0] = hf[0];
            s_h[1] = hf[1];
            s_h[2] = hf[2];
            if (round < (uint32_t)kMaxPeelModels) {  // (d_models holds kMaxPeelModels refits; enqueue_groups rejects more)
                models[round * 3 + 0] = hf[0];
                models[round * 3 + 1] = hf[1];
                models[round * 3 + 2] = hf[2];
            }
        }
    }
    __syncthreads();
    const float fx = s_h[0], fy = s_h[1], fz = s_h[2];
#ifdef LR_PEEL_TIMING
    tm[4] = wall_clock64();
#endif

    // -- the round's verdict per line (estimator.h:122-135): inlier -> id of the round; near miss -> garbage (out of the
    // game, final id -1); the others go on, in order, into the next round's table
    uint32_t n_next = 0, n_gone = 0;
    for (uint32_t i0 = 0; i0 < n; i0 += kWG) {
        const uint32_t i = i0 + threadIdx.x;
        bool stay = false, gone = false;
        if (i < n) {
            if (i0 != 0) {
                r_ax = cur.ax[i]; r_ay = cur.ay[i]; r_dx = cur.dx[i]; r_dy = cur.dy[i]; r_len = cur.len[i];
                r_hx = cur.hx[i]; r_hy = cur.hy[i]; r_hz = cur.hz[i]; r_orig = cur.orig[i];
            }
            const float e = pencil_error(r_ax, r_ay, r_dx, r_dy, fx, fy, fz);
            if (e < tol) {
                lines[r_orig].group_id = (int)round;
                gone = true;
            } else if (e >= tol && e < garbage_tol) {
                gone = true;
            } else {
                stay = true;  // (a NaN error stays, as in the reference: both comparisons are false)
            }
        }
        uint32_t tot, tg;
        const uint32_t r = block_rank(stay, s_cnt, tot);
        if (stay) {
            const uint32_t o = n_next + r;
            nxt.ax[o] = r_ax; nxt.ay[o] = r_ay; nxt.dx[o] = r_dx; nxt.dy[o] = r_dy;
            nxt.len[o] = r_len; nxt.hx[o] = r_hx; nxt.hy[o] = r_hy; nxt.hz[o] = r_hz;
            nxt.orig[o] = r_orig;
        }
        (void)block_rank(gone, s_cnt, tg);
        n_next += tot;
        n_gone += tg;
    }
#ifdef LR_PEEL_TIMING
    if (threadIdx.x == 0) {
        tm[5] = wall_clock64();
        for (int q = 0; q < 5 && round < 4u; ++q) models[16 + round * 5 + q] = (float)(tm[q + 1] - tm[q]) * 0.01f;  // us (100 MHz clock)
    }
#endif
    if (threadIdx.x == 0) {
        gctl[kGcActive] = n_next;
        gctl[kGcRemaining] = gctl[kGcRemaining] - n_gone;
        gctl[kGcRound] = round + 1u;
    }
}

// The launch: a peeling round, and -- the frame's LAST round only (gather_dst != nullptr) -- the frame's results into the
// page-locked block the host reads after its one wait (round 5: that was a launch of its own behind the last round).  Layout
// (context.h kResHeaderBytes = 256): words 0..7 the stage counts, words 8..15 the peeling control block, words 16..55 the
// refit models, from byte 256 on the grouped lines (as many as there are, up to the block's capacity): only the lines that
// exist cross the link.  The last round enqueued runs whether the peeling is over by then or not.
__global__ __launch_bounds__(kWG) void peel_kernel(PencilTable cur, PencilTable nxt, PencilTable all,
                                                   unsigned long long* __restrict__ best_slots, uint64_t seed,
                                                   float tol, float garbage_tol, int max_models,
                                                   uint32_t* __restrict__ gctl, float4* __restrict__ stage_g,
                                                   LineSegment* __restrict__ lines, float* __restrict__ models,
                                                   const uint32_t* __restrict__ counts, uint32_t cap_lines,
                                                   uint32_t* __restrict__ gather_dst) {
    peel_body(cur, nxt, all, best_slots, seed, tol, garbage_tol, max_models, gctl, stage_g, lines, models);
    if (gather_dst == nullptr) return;
    __syncthreads();  // (the round's own stores -- group ids, control words, the refit model -- are visible to the workgroup)
    const uint32_t t = threadIdx.x;
    const uint32_t* gw = gctl;
    if (t < 8) gather_dst[t] = counts[t];
    else if (t < 16) gather_dst[t] = gw[t - 8];
    else if (t < 56) gather_dst[t] = reinterpret_cast<const uint32_t*>(models)[t - 16];
    const uint32_t n = min(gw[kGcLines], cap_lines) * (uint32_t)(sizeof(LineSegment) / 4);
    const uint32_t* lw = reinterpret_cast<const uint32_t*>(lines);
    for (uint32_t i = t; i < n; i += kWG) gather_dst[64 + i] = lw[i];
}

}  // namespace

int launch_filter_lines(const LineSegment* raw, const uint32_t* d_n_raw, uint32_t raw_cap, float min_length,
                        LineSegment* out, uint32_t* gctl, float* gnorm, const PencilTable* all, const PencilTable* round0,
                        hipStream_t s) {
    const bool with_model = all != nullptr && round0 != nullptr;
    hipLaunchKernelGGL(filter_lines_kernel, dim3(1), dim3(kWG), 0, s, raw, d_n_raw, raw_cap, min_length, out, gctl, gnorm,
                       with_model ? *all : PencilTable{}, with_model ? *round0 : PencilTable{}, with_model ? 1u : 0u);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_lines_bbox(LineSegment* lines, uint32_t n, uint32_t* gctl, float* gnorm, hipStream_t s) {
    hipLaunchKernelGGL(lines_bbox_kernel, dim3(1), dim3(kWG), 0, s, lines, n, gctl, gnorm);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_pencil_model(const LineSegment* lines, const uint32_t* gctl, const float* gnorm, PencilTable all,
                        PencilTable round0, uint32_t line_cap, hipStream_t s) {
    if (line_cap == 0) return 0;
    hipLaunchKernelGGL(pencil_model_kernel, dim3((line_cap + 255) / 256), dim3(256), 0, s, lines, gctl, gnorm, all, round0);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_peel(PencilTable cur, PencilTable nxt, PencilTable all, unsigned long long* best_slots, uint64_t seed,
                float tol, float garbage_tol, int max_models, uint32_t* gctl, float* stage4, LineSegment* lines,
                float* models, const uint32_t* counts, uint32_t cap_lines, void* gather_block, hipStream_t s) {
    hipLaunchKernelGGL(peel_kernel, dim3(1), dim3(kWG), 0, s, cur, nxt, all, best_slots, seed, tol, garbage_tol,
                       max_models, gctl, reinterpret_cast<float4*>(stage4), lines, models, counts, cap_lines,
                       static_cast<uint32_t*>(gather_block));
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
