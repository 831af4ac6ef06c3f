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
#include <cstdlib>

#include "common.h"

namespace lramd {
namespace {

__device__ inline float wave_tree(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v = v + __shfl_xor(v, off);
    return v;
}

// ---- inlier test --------------------------------------------------------------------------------------------------
// The canonical test is   err = 1 - |u . d| < tol   with u = v / |v| from a correctly rounded square root and two
// correctly rounded divisions (geometry.cpp:214-229): some forty vector instructions per (hypothesis, line), which is
// what bounds these kernels.  Only the DECISION enters a score, so it is taken from a cheap estimate whenever that
// is safe: err' = 1 - |v . d| * rsq(|v|^2) differs from err by at most 13 units of 2^-24 (the error analysis is in
// DESIGN.md §5c), i.e. by less than 8e-7, and if err' is further than kErrBand = 3e-6 from tol the two agree.  Inside
// the band -- and whenever the estimate is not a number -- the canonical expression decides.  The results are
// therefore bit-identical to the canonical ones.
constexpr float kErrBand = 3e-6f;

__device__ __forceinline__ bool inlier_exact(float vx, float vy, float dx, float dy, float tol) {
    const float nn = vx * vx + vy * vy;
    const float nrm = sqrtf(nn);
    const float ux = vx / nrm, uy = vy / nrm;
    const float inc = fabsf(ux * dx + uy * dy);
    const float err = -inc + 1.0f;
    return err < tol;
}

__device__ __forceinline__ bool inlier_test(float vx, float vy, float dx, float dy, float tol) {
    const float nn = __builtin_fmaf(vx, vx, vy * vy);
    const float dot = __builtin_fmaf(vx, dx, vy * dy);
    const float e = 1.0f - fabsf(dot) * __builtin_amdgcn_rsqf(nn);
    bool in = e < tol - kErrBand;
    // not sure: inside the band, not a number, or |v|^2 so small that the reciprocal square root (which flushes
    // denormals) and the canonical 0 / 0 could part ways
    const bool sure = (in || e > tol + kErrBand) && nn > 1e-30f;
    if (__builtin_expect(!sure, 0)) in = inlier_exact(vx, vy, dx, dy, tol);
    return in;
}

// A hypothesis as the scoring loop needs it: v = (cx, cy) - sub * anchor  (sub = 0 for an ideal point, whose v is
// the point itself: geometry.cpp:218-223)
struct Hyp {
    float cx, cy, sub;
    bool valid;
};
__device__ __forceinline__ Hyp make_hyp(const PencilSoA& m, uint32_t a, uint32_t b, float degeneracy_tol) {
    const float hax = m.hx[a], hay = m.hy[a], haz = m.hz[a];
    const float hbx = m.hx[b], hby = m.hy[b], hbz = m.hz[b];
    const float ex = hax - hbx, ey = hay - hby, ez = haz - hbz;
    const float dist = sqrtf((ex * ex + ey * ey) + ez * ez);
    Hyp h;
    h.valid = dist > degeneracy_tol;  // sample_check (line_pencil.cpp:89-98)
    // fit: h_a x h_b (line_pencil.cpp:101-108)
    const float px = hay * hbz - haz * hby;
    const float py = haz * hbx - hax * hbz;
    const float pz = hax * hby - hay * hbx;
    const bool ideal = fabsf(pz) < kEps;
    h.cx = ideal ? px : px / pz;
    h.cy = ideal ? py : py / pz;
    h.sub = ideal ? 0.f : 1.f;
    return h;
}
// v for line i: exactly (cx - ax, cy - ay) or (cx, cy)
__device__ __forceinline__ void hyp_v(const Hyp& h, float ax, float ay, float& vx, float& vy) {
    vx = h.sub != 0.f ? h.cx - ax : h.cx;
    vy = h.sub != 0.f ? h.cy - ay : h.cy;
}

// The estimate of the inlier test without a branch, a division or a square root.  With v = c - sub * anchor (two FMAs:
// sub is 0 or 1, fma(-1, a, c) is the rounded difference and fma(-0, a, c) is c itself), nn = |v|^2 and dot = v . d, the
// canonical error err = 1 - |dot| / |v| is below tol iff dot^2 > (1 - tol)^2 nn.  The estimate tests that inequality
// with a safety band around the threshold: `in` where dot^2 > k_in nn with k_in = (1 - tol + band)^2, "surely out"
// where dot^2 < k_out nn with k_out = (1 - tol - band)^2, and everything else -- inside the band, not a number,
// |v| = 0, or |v|^2 so small that the products could underflow -- is `unsure` and decided by the canonical expression.
// The six products and sums carry a relative error below 2^-21 together, i.e. less than 5e-7 on |dot| / |v| <= 1, and
// the canonical float expression is within 13 x 2^-24 = 8e-7 of the real quotient: with
// band = 3e-6 a sure decision is the canonical decision.  Scores and counts are therefore bit-identical.
constexpr float kTiny = 1e-30f;
struct EstConst {
    float k_in, k_out;
};
__device__ __forceinline__ EstConst est_const(float tol) {
    const float a = (1.0f - tol) + kErrBand, b = (1.0f - tol) - kErrBand;
    return EstConst{a * a, b * b};
}
// Evaluated for TWO hypotheses at once in packed fp32 (v_pk_fma_f32 / v_pk_mul_f32: both halves are IEEE operations, the
// masks are what a scalar evaluation would give): nine packed instructions instead of eighteen.  Thresholds carry a floor
// of kTiny on either side: for |v|^2 below it neither comparison can hold (dot^2 <= |v|^2), so such a line is unsure without
// a comparison of its own.  The lane masks come straight from the comparisons (v_cmp writes an SGPR pair) and are combined
// on the scalar unit.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void inlier_estimate2(const f2 cx, const f2 cy, const f2 nsub, float ax, float ay, float dx, float dy,
                                                 const EstConst& K, uint64_t& m_in0, uint64_t& m_un0, uint64_t& m_in1,
                                                 uint64_t& m_un1) {
    const f2 vx = __builtin_elementwise_fma(nsub, (f2){ax, ax}, cx), vy = __builtin_elementwise_fma(nsub, (f2){ay, ay}, cy);
    const f2 nn = __builtin_elementwise_fma(vx, vx, vy * vy);
    const f2 dot = __builtin_elementwise_fma(vx, (f2){dx, dx}, vy * (f2){dy, dy});
    const f2 dd = dot * dot;
    const f2 t_in = __builtin_elementwise_fma(nn, (f2){K.k_in, K.k_in}, (f2){kTiny, kTiny});
    const f2 t_out = __builtin_elementwise_fma(nn, (f2){K.k_out, K.k_out}, (f2){-kTiny, -kTiny});
    m_in0 = __builtin_amdgcn_fcmpf(dd.x, t_in.x, 2);
    m_in1 = __builtin_amdgcn_fcmpf(dd.y, t_in.y, 2);
    m_un0 = ~(m_in0 | __builtin_amdgcn_fcmpf(dd.x, t_out.x, 4));
    m_un1 = ~(m_in1 | __builtin_amdgcn_fcmpf(dd.y, t_out.y, 4));
}

// One wavefront scores kH hypotheses at a time, and the four wavefronts of a workgroup share one pass over the line
// table: it is staged through LDS in chunks of kScoreChunk lines (two buffers, one barrier per chunk), so 4 x kH
// hypotheses cost one read of the table from L2 (it used to be read once per wavefront: at N = 20 000 and 100 000
// hypotheses that was 5 GB per solve).  Each lane tests its lines against all kH hypotheses (whose parameters are
// wave-uniform) in one straight-line block; the rare lines for which an estimate is not sure are decided by the
// canonical expression afterwards.  The score of each hypothesis is the canonical tree T() over the lines (lane-strided
// partial sums in ascending line order, then the xor butterfly), exactly as when a wavefront owned a single hypothesis.
// gctl != nullptr: line count and round come from the device (peeling rounds enqueued blindly, kernels_groups.hip);
// the kernel then leaves at once when the peeling is over.
constexpr int kScoreChunk = 512;
constexpr uint32_t kBestSlots = (uint32_t)kRansacBestSlots;
template <int kH>
__global__ __launch_bounds__(256) void ransac_score_kernel(PencilSoA m, uint32_t n_host, float tol, float degeneracy_tol,
                                                           uint32_t n_iter, uint64_t seed, uint32_t round_host,
                                                           const uint32_t* __restrict__ gctl, int max_models,
                                                           unsigned long long* __restrict__ best_slots) {
    __shared__ float s_tab[2][5][kScoreChunk];
    __shared__ unsigned long long s_key[4];
    const int lane = threadIdx.x & 63;
    const uint32_t hyp0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kH;
    if (blockIdx.x * 4u * kH >= n_iter) return;  // (whole workgroup)
    uint32_t n = n_host, round = round_host;
    if (gctl) {
        n = gctl[kGcActive];
        round = gctl[kGcRound];
        if (gctl[kGcRemaining] < 2u || round >= (uint32_t)max_models) return;
    }
    float cx[kH], cy[kH], nsub[kH];
    bool valid[kH];
#pragma unroll
    for (int j = 0; j < kH; ++j) {
        uint32_t a, b;
        sample_pair(seed, round, min(hyp0 + (uint32_t)j, n_iter - 1u), n, a, b);
        const Hyp h = make_hyp(m, a, b, degeneracy_tol);
        cx[j] = h.cx;
        cy[j] = h.cy;
        nsub[j] = -h.sub;
        valid[j] = h.valid;
    }
    const EstConst K = est_const(tol);
    float acc[kH];
#pragma unroll
    for (int j = 0; j < kH; ++j) acc[j] = 0.f;
    auto stage = [&](int buf, uint32_t c0) {
        for (uint32_t t = threadIdx.x; t < (uint32_t)kScoreChunk; t += 256) {
            const uint32_t i = c0 + t;
            if (i < n) {
                s_tab[buf][0][t] = m.ax[i];
                s_tab[buf][1][t] = m.ay[i];
                s_tab[buf][2][t] = m.dx[i];
                s_tab[buf][3][t] = m.dy[i];
                s_tab[buf][4][t] = m.len[i];
            }
        }
    };
    stage(0, 0u);
    int buf = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += kScoreChunk, buf ^= 1) {
        __syncthreads();  // this chunk is in LDS; the other buffer is no longer being read
        if (c0 + kScoreChunk < n) stage(buf ^ 1, c0 + kScoreChunk);
        const uint32_t cn = min((uint32_t)kScoreChunk, n - c0);
        for (uint32_t t = lane; t < cn; t += 64) {
            const float ax = s_tab[buf][0][t], ay = s_tab[buf][1][t], dx = s_tab[buf][2][t], dy = s_tab[buf][3][t];
            const float len = s_tab[buf][4][t];
            uint64_t any_unsure = 0ull, mu[kH];
            static_assert(kH % 2 == 0, "hypotheses are scored in pairs");
#pragma unroll
            for (int j = 0; j < kH; j += 2) {
                uint64_t mi0, mi1;
                inlier_estimate2((f2){cx[j], cx[j + 1]}, (f2){cy[j], cy[j + 1]}, (f2){nsub[j], nsub[j + 1]}, ax, ay, dx, dy, K, mi0,
                                 mu[j], mi1, mu[j + 1]);
                // (an unsure line adds nothing here; the canonical test below adds it if it is an inlier)
                acc[j] = acc[j] + (__builtin_amdgcn_inverse_ballot_w64(mi0 & ~mu[j]) ? len : 0.0f);
                acc[j + 1] = acc[j + 1] + (__builtin_amdgcn_inverse_ballot_w64(mi1 & ~mu[j + 1]) ? len : 0.0f);
                any_unsure |= mu[j] | mu[j + 1];
            }
            if (__builtin_expect(any_unsure != 0ull, 0)) {  // wave-uniform, rare
#pragma unroll
                for (int j = 0; j < kH; ++j)
                    if (__builtin_amdgcn_inverse_ballot_w64(mu[j])) {
                        const float vx = __builtin_fmaf(nsub[j], ax, cx[j]), vy = __builtin_fmaf(nsub[j], ay, cy[j]);
                        acc[j] = acc[j] + (inlier_exact(vx, vy, dx, dy, tol) ? len : 0.0f);
                    }
            }
        }
    }
    // The scores themselves are never stored: all that estimator.h:62-70 keeps of an iteration is whether it is the first
    // strictly best one.  Positive scores order like their bit patterns, so "highest score, lowest iteration among equals" is
    // the maximum of score bits : 32 | ~iteration : 32, taken per wavefront, per workgroup, and with one 64-bit atomic into
    // one of kBestSlots device words (round 3 wrote 10 000 scores and scanned them in a second launch).
    unsigned long long key = 0ull;
#pragma unroll
    for (int j = 0; j < kH; ++j) {
        const float score = wave_tree(acc[j]);
        const uint32_t it = hyp0 + (uint32_t)j;
        if (valid[j] && it < n_iter && score > 0.0f) {
            const unsigned long long k = ((unsigned long long)__float_as_uint(score) << 32) | (unsigned long long)(0xFFFFFFFFu - it);
            key = k > key ? k : key;
        }
    }
    if (lane == 0) s_key[threadIdx.x >> 6] = key;
    __syncthreads();
    if (threadIdx.x != 0) return;
    for (int wv = 1; wv < 4; ++wv) key = s_key[wv] > key ? s_key[wv] : key;
    // (fire and forget: no returned value is waited for -- a device-scope atomic is a microsecond's round trip, and a
    // workgroup that waits for two of them at its end made the 8 us scoring launches of a frame 3 us longer; twenty
    // same-address atomics per slot are nothing)
    if (key != 0ull) (void)__hip_atomic_fetch_max(&best_slots[blockIdx.x % kBestSlots], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// lr_ransac_best: the slots' maximum goes straight into page-locked host memory (score bits, iteration; -1 if nothing
// scored), and the slots are cleared for the next solve.  (A last-workgroup-done finish inside the scoring launch was built
// first: its tickets -- device-scope atomics WITH a returned value, one or two per workgroup -- cost more than this launch:
// 62 us a solve of 10 000 hypotheses against 56 before, 1.02 ms against 0.94 at 100 000.)
__global__ __launch_bounds__(64) void ransac_best_readout_kernel(unsigned long long* __restrict__ best_slots,
                                                                 uint32_t* __restrict__ host_best) {
    unsigned long long key = threadIdx.x < kBestSlots ? best_slots[threadIdx.x] : 0ull;
    if (threadIdx.x < kBestSlots) best_slots[threadIdx.x] = 0ull;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long o = ((unsigned long long)(uint32_t)__shfl_xor((int)(uint32_t)(key >> 32), off) << 32) |
                                     (uint32_t)__shfl_xor((int)(uint32_t)key, off);
        key = o > key ? o : key;
    }
    if (threadIdx.x == 0) {
        host_best[0] = (uint32_t)(key >> 32);
        host_best[1] = key ? 0xFFFFFFFFu - (uint32_t)key : 0xFFFFFFFFu;
    }
}

// ---- PROSAC support (reference prosac.h, line_pencil.cpp:47-86; opt-in) ----------

// Inlier COUNT (prosac.h:208-210) of explicit two-line samples over the quality-sorted line table.
// kH hypotheses per wavefront, the table staged through LDS for the workgroup's four wavefronts (as ransac_score_kernel);
// the count is an integer, so its reduction order is immaterial.
template <int kH>
__global__ __launch_bounds__(256) void prosac_count_kernel(PencilSoA m, uint32_t n, float tol, float degeneracy_tol,
                                                           const uint32_t* __restrict__ sa,
                                                           const uint32_t* __restrict__ sb, uint32_t n_hyp,
                                                           uint32_t* __restrict__ counts) {
    __shared__ float s_tab[2][4][kScoreChunk];
    const int lane = threadIdx.x & 63;
    const uint32_t hyp0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * kH;
    if (blockIdx.x * 4u * kH >= n_hyp) return;  // (whole workgroup)
    float cx[kH], cy[kH], nsub[kH];
    bool valid[kH];
#pragma unroll
    for (int j = 0; j < kH; ++j) {
        const uint32_t h = min(hyp0 + (uint32_t)j, n_hyp - 1u);
        const Hyp hy = make_hyp(m, sa[h], sb[h], degeneracy_tol);
        cx[j] = hy.cx;
        cy[j] = hy.cy;
        nsub[j] = -hy.sub;
        valid[j] = hy.valid;
    }
    const EstConst K = est_const(tol);
    uint32_t cnt[kH];
#pragma unroll
    for (int j = 0; j < kH; ++j) cnt[j] = 0u;
    auto stage = [&](int buf, uint32_t c0) {
        for (uint32_t t = threadIdx.x; t < (uint32_t)kScoreChunk; t += 256) {
            const uint32_t i = c0 + t;
            if (i < n) {
                s_tab[buf][0][t] = m.ax[i];
                s_tab[buf][1][t] = m.ay[i];
                s_tab[buf][2][t] = m.dx[i];
                s_tab[buf][3][t] = m.dy[i];
            }
        }
    };
    stage(0, 0u);
    int buf = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += kScoreChunk, buf ^= 1) {
        __syncthreads();
        if (c0 + kScoreChunk < n) stage(buf ^ 1, c0 + kScoreChunk);
        const uint32_t cn = min((uint32_t)kScoreChunk, n - c0);
        for (uint32_t t = lane; t < cn; t += 64) {
            const float ax = s_tab[buf][0][t], ay = s_tab[buf][1][t], dx = s_tab[buf][2][t], dy = s_tab[buf][3][t];
            uint64_t any_unsure = 0ull, mu[kH];
#pragma unroll
            for (int j = 0; j < kH; j += 2) {
                uint64_t mi0, mi1;
                inlier_estimate2((f2){cx[j], cx[j + 1]}, (f2){cy[j], cy[j + 1]}, (f2){nsub[j], nsub[j + 1]}, ax, ay, dx, dy, K, mi0,
                                 mu[j], mi1, mu[j + 1]);
                cnt[j] += __builtin_amdgcn_inverse_ballot_w64(mi0 & ~mu[j]) ? 1u : 0u;
                cnt[j + 1] += __builtin_amdgcn_inverse_ballot_w64(mi1 & ~mu[j + 1]) ? 1u : 0u;
                any_unsure |= mu[j] | mu[j + 1];
            }
            if (__builtin_expect(any_unsure != 0ull, 0)) {  // wave-uniform, rare
#pragma unroll
                for (int j = 0; j < kH; ++j)
                    if (__builtin_amdgcn_inverse_ballot_w64(mu[j])) {
                        const float vx = __builtin_fmaf(nsub[j], ax, cx[j]), vy = __builtin_fmaf(nsub[j], ay, cy[j]);
                        cnt[j] += inlier_exact(vx, vy, dx, dy, tol) ? 1u : 0u;
                    }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < kH; ++j) {
        uint32_t c = cnt[j];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) c += (uint32_t)__shfl_xor((int)c, off);
        // sample_check failed: the reference skips the iteration
        if (lane == 0 && hyp0 + (uint32_t)j < n_hyp) counts[hyp0 + j] = valid[j] ? c : 0xFFFFFFFFu;
    }
}

// Inlier flags of ONE hypothesis over the whole table (prosac.h:212-222: when a sample is a new best, its inliers are
// needed line by line for the maximality test): canonical test, one line per thread.
__global__ __launch_bounds__(256) void prosac_flags_kernel(PencilSoA m, uint32_t n, float px, float py, float pz, float tol,
                                                           uint8_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float vx, vy;
    if (fabsf(pz) < kEps) {
        vx = px;
        vy = py;
    } else {
        vx = px / pz - m.ax[i];
        vy = py / pz - m.ay[i];
    }
    flags[i] = inlier_exact(vx, vy, m.dx[i], m.dy[i], tol) ? 1 : 0;
}

// The iterations of a speculative chunk at which PROSAC finds a new best hypothesis (prosac.h:208-222) follow from the
// inlier counts alone: iteration j is one iff its count exceeds every earlier valid count of the chunk and the best count
// the chunk started with.  One workgroup lists them in order (at most `cap`; *n_rec is their true number), so that the
// inlier flags the host needs at each of them can be produced before it looks at the chunk -- one wait per chunk
// instead of one per new best.
__global__ __launch_bounds__(1024) void prosac_records_kernel(const uint32_t* __restrict__ counts, uint32_t n_hyp, uint32_t best_in,
                                                              uint32_t* __restrict__ rec /* [0] = n_rec, [1..cap] = iterations */,
                                                              uint32_t cap) {
    __shared__ uint32_t s_max[1024];
    __shared__ uint32_t s_cnt[1024];
    const uint32_t L = (n_hyp + 1023u) / 1024u;
    const uint32_t i0 = threadIdx.x * L, i1 = min(n_hyp, i0 + L);
    uint32_t mx = 0;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t c = counts[i];
        if (c != 0xFFFFFFFFu) mx = max(mx, c);
    }
    s_max[threadIdx.x] = mx;
    __syncthreads();
    // best count before this thread's run (a serial pass over 1024 words per thread would do as well; this is a tree)
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t o = threadIdx.x >= (unsigned)off ? s_max[threadIdx.x - off] : 0u;
        __syncthreads();
        s_max[threadIdx.x] = max(s_max[threadIdx.x], o);
        __syncthreads();
    }
    uint32_t run = max(best_in, threadIdx.x ? s_max[threadIdx.x - 1] : 0u);
    uint32_t n = 0;
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t c = counts[i];
        if (c != 0xFFFFFFFFu && c > run) {
            run = c;
            ++n;
        }
    }
    s_cnt[threadIdx.x] = n;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t o = threadIdx.x >= (unsigned)off ? s_cnt[threadIdx.x - off] : 0u;
        __syncthreads();
        s_cnt[threadIdx.x] += o;
        __syncthreads();
    }
    uint32_t pos = s_cnt[threadIdx.x] - n;
    run = max(best_in, threadIdx.x ? s_max[threadIdx.x - 1] : 0u);
    for (uint32_t i = i0; i < i1; ++i) {
        const uint32_t c = counts[i];
        if (c != 0xFFFFFFFFu && c > run) {
            run = c;
            if (pos < cap) rec[1 + pos] = i;
            ++pos;
        }
    }
    if (threadIdx.x == 1023) rec[0] = s_cnt[1023];
}

// Inlier flags (canonical test, as prosac_flags_kernel) of the listed iterations' hypotheses h_a x h_b, one row of n
// bytes per listed iteration.
__global__ __launch_bounds__(256) void prosac_record_flags_kernel(PencilSoA m, uint32_t n, const uint32_t* __restrict__ sa,
                                                                  const uint32_t* __restrict__ sb,
                                                                  const uint32_t* __restrict__ rec, uint32_t cap, float tol,
                                                                  uint8_t* __restrict__ flags) {
    const uint32_t r = blockIdx.y;
    if (r >= min(rec[0], cap)) return;
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t j = rec[1 + r];
    const uint32_t a = sa[j], b = sb[j];
    const float hax = m.hx[a], hay = m.hy[a], haz = m.hz[a];
    const float hbx = m.hx[b], hby = m.hy[b], hbz = m.hz[b];
    const float px = hay * hbz - haz * hby;  // line_pencil.cpp:101-108
    const float py = haz * hbx - hax * hbz;
    const float pz = hax * hby - hay * hbx;
    float vx, vy;
    if (fabsf(pz) < kEps) {
        vx = px;
        vy = py;
    } else {
        vx = px / pz - m.ax[i];
        vy = py / pz - m.ay[i];
    }
    flags[(size_t)r * n + i] = inlier_exact(vx, vy, m.dx[i], m.dy[i], tol) ? 1 : 0;
}

// Hough votes of get_weights on the unit hemisphere: ht x ht accumulator in LDS, 64-bit integer atomics
// (votes in 2^-20 fixed point, so the result does not depend on arrival order), then the first maximum
// in column-major order.  Single workgroup: 20 000 votes are nothing.
constexpr int kHtMax = 65;
__global__ __launch_bounds__(1024) void ht_votes_kernel(PencilSoA m, const int32_t* __restrict__ pa,
                                                        const int32_t* __restrict__ pb, int n_pairs, int ht,
                                                        float* __restrict__ peak /* 3 floats */) {
    __shared__ unsigned long long acc[kHtMax * kHtMax];
    __shared__ unsigned long long s_best[16];
    __shared__ int s_pos[16];
    const float k = floorf(ht / 2.f), k1 = k - 1;
    for (int i = threadIdx.x; i < ht * ht; i += 1024) acc[i] = 0ull;
    __syncthreads();
    for (int i = threadIdx.x; i < n_pairs; i += 1024) {
        const int a = pa[i], b = pb[i];
        float x = m.hy[a] * m.hz[b] - m.hz[a] * m.hy[b];
        float y = m.hz[a] * m.hx[b] - m.hx[a] * m.hz[b];
        float z = m.hx[a] * m.hy[b] - m.hy[a] * m.hx[b];
        if (fabsf(x) < 0.0001f && fabsf(y) < 0.0001f && fabsf(z) < 0.0001f) continue;
        const float zz = (x * x + y * y) + z * z;
        if (zz > 0.0f) {
            const float nn = sqrtf(zz);
            x = x / nn;
            y = y / nn;
            z = z / nn;
        }
        if (z < 0.f) {
            x = -x;
            y = -y;
        }
        const int u = (int)roundf(k1 * x + k);
        const int v = (int)roundf(k1 * y + k);
        const float vote = m.len[a] + m.len[b];
        atomicAdd(&acc[u * ht + v], (unsigned long long)(vote * 1048576.0f + 0.5f));
    }
    __syncthreads();
    // first strict maximum in column-major order == maximum value, lowest column-major position
    unsigned long long bv = 0ull;
    int bp = 0x7FFFFFFF;
    for (int i = threadIdx.x; i < ht * ht; i += 1024) {
        const int u = i / ht, v = i - u * ht;
        const int pos = v * ht + u;
        const unsigned long long val = acc[i];
        if (val > bv || (val == bv && pos < bp)) {
            bv = val;
            bp = pos;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned long long ov = __shfl_xor(bv, off);
        const int op = __shfl_xor(bp, off);
        if (ov > bv || (ov == bv && op < bp)) {
            bv = ov;
            bp = op;
        }
    }
    if ((threadIdx.x & 63) == 0) {
        s_best[threadIdx.x >> 6] = bv;
        s_pos[threadIdx.x >> 6] = bp;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w)
            if (s_best[w] > bv || (s_best[w] == bv && s_pos[w] < bp)) {
                bv = s_best[w];
                bp = s_pos[w];
            }
        const int max_v = bp / ht, max_u = bp - max_v * ht;
        float p0 = (max_u - k) / k1, p1 = (max_v - k) / k1;
        const float pn = sqrtf((p0 * p0 + p1 * p1) + 0.f * 0.f);
        if (pn > 1.f) {
            p0 = p0 / pn;
            p1 = p1 / pn;
        }
        peak[0] = p0;
        peak[1] = p1;
        peak[2] = sqrtf(1.f - (p0 * p0 + p1 * p1));
    }
}

__global__ __launch_bounds__(256) void ht_weights_kernel(PencilSoA m, uint32_t n, const float* __restrict__ peak,
                                                         float* __restrict__ weights) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float px = peak[0], py = peak[1], pz = peak[2];
    float vx, vy;
    if (fabsf(pz) < kEps) {
        vx = px;
        vy = py;
    } else {
        vx = px / pz - m.ax[i];
        vy = py / pz - m.ay[i];
    }
    const float nrm = sqrtf(vx * vx + vy * vy);
    const float ux = vx / nrm, uy = vy / nrm;
    const float inc = fabsf(ux * m.dx[i] + uy * m.dy[i]);
    const float i2 = inc * inc;
    weights[i] = i2 * i2;
}

// ---- refine: pairwise merge test (reference line_detector.cpp:353-400) ---------------------------------
// One thread per (i, j > i) pair inside a 64 x 64 tile of the pair matrix; segment data of the tile's rows and
// columns is staged in LDS.  Emits the edges (i, j) of the merge graph; the (tiny) graph walk and the merges
// stay on the host.  Same float operations, in the same order, as the host loop in vp_host.cpp.
struct RefineSeg {
    float x1, y1, x2, y2, dx, dy, len;  // d = unit direction, n = (-dy, dx)
};

__global__ __launch_bounds__(256) void refine_pairs_kernel(const RefineSeg* __restrict__ seg, uint32_t n,
                                                           uint2* __restrict__ edges, uint32_t* __restrict__ n_edges,
                                                           uint32_t cap) {
    __shared__ RefineSeg si[64], sj[64];
    const uint32_t bi = blockIdx.y, bj = blockIdx.x;
    if (bj < bi) return;  // only the upper triangle (whole block)
    const uint32_t i0 = bi * 64, j0 = bj * 64;
    for (uint32_t t = threadIdx.x; t < 128; t += 256) {
        const uint32_t idx = (t < 64) ? i0 + t : j0 + (t - 64);
        RefineSeg v{0, 0, 0, 0, 0, 0, 1};
        if (idx < n) v = seg[idx];
        if (t < 64) si[t] = v; else sj[t - 64] = v;
    }
    __syncthreads();
    for (uint32_t p = threadIdx.x; p < 64 * 64; p += 256) {
        const uint32_t a = p >> 6, b = p & 63;
        const uint32_t i = i0 + a, j = j0 + b;
        if (i >= n || j >= n || j <= i) continue;
        const RefineSeg A = si[a], B = sj[b];
        if (fabsf(A.dx * B.dx + A.dy * B.dy) < 0.99) continue;  // cos(max angular difference); double literal as the reference
        const bool i_short = A.len < B.len;
        const RefineSeg R = i_short ? B : A;  // frame of the longer one
        const RefineSeg O = i_short ? A : B;
        const float nx = -R.dy, ny = R.dx;
        const float ax = O.x1 - R.x1, ay = O.y1 - R.y1, bx = O.x2 - R.x1, by = O.y2 - R.y1;
        const float w00 = (ax * R.dx + ay * R.dy) / R.len, w01 = (ax * nx + ay * ny) / R.len;
        const float w10 = (bx * R.dx + by * R.dy) / R.len, w11 = (bx * nx + by * ny) / R.len;
        if (fmaxf(fabsf(w01), fabsf(w11)) < 0.02) {
            const bool any_gt = (w00 > -0.5) || (w10 > -0.5);
            const bool any_lt = (w00 < 1.5) || (w10 < 1.5);
            if (any_gt && any_lt) {
                const uint32_t slot = atomicAdd(n_edges, 1u);
                if (slot < cap) edges[slot] = make_uint2(i, j);
            }
        }
    }
}

// ---- diamond-space (cascaded Hough) accumulator, opt-in (reference cht.h:13-24, cht.cpp: an uncompiled sketch) --
// One WAVEFRONT per line: the four corner points of the line's polyline are wave-uniform arithmetic, and the cells of
// each of its three segments (up to d of them: cht.cpp:163-197 steps along the longer axis) are voted for by the lanes
// side by side.  Each workgroup keeps a d x d accumulator of 32-bit fixed-point votes in LDS (d <= 128: 64 KB) and
// flushes it into the global 64-bit accumulator with integer atomics, so the result does not depend on scheduling --
// and votes can be TAKEN BACK exactly (cht.h:18: "the weights can be negative (so lines can be removed!)"): the
// peeling loop subtracts the lines a round has removed instead of accumulating the rest again.
// Formulas and rasterisation: see the oracle's cht_accumulate.
__device__ inline float sgn1(float x) { return x >= 0.f ? 1.f : -1.f; }
constexpr int kChtMax = 128;
constexpr int kChtLinesPerBlock = 512;  // keeps every LDS cell below 2^32 (<= 6 votes of < 2^17 per line)

// idx == nullptr: lines 0..n-1 of the table; else lines idx[0..n-1].  kSub: the votes are taken back.
template <bool kSub>
__global__ __launch_bounds__(256) void cht_votes_kernel(PencilSoA m, const uint32_t* __restrict__ idx, uint32_t n, int d,
                                                        unsigned long long* __restrict__ acc,
                                                        unsigned long long* __restrict__ n_votes) {
    extern __shared__ uint32_t s_acc[];
    for (int i = threadIdx.x; i < d * d; i += 256) s_acc[i] = 0u;
    __syncthreads();
    const float sc = (float)(d - 1);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const uint32_t lo = blockIdx.x * kChtLinesPerBlock;
    const uint32_t hi = min(n, lo + kChtLinesPerBlock);
    uint32_t cast = 0;
    for (uint32_t li = lo + wv; li < hi; li += 4) {
        const uint32_t i = idx ? idx[li] : li;
        const float a = m.hx[i], b = m.hy[i], c = m.hz[i];
        const float al = sgn1(a * b), be = sgn1(b * c), ga = sgn1(a * c);
        const float d1 = c + ga * a, d2 = c + be * b, d3 = a + al * b;
        float px[4], py[4];
        bool ok[4];
        ok[0] = ok[3] = d1 != 0.f;
        ok[1] = d2 != 0.f;
        ok[2] = d3 != 0.f;
        px[0] = al * a / d1;   py[0] = -al * c / d1;
        px[1] = b / d2;        py[1] = 0.f;
        px[2] = 0.f;           py[2] = b / d3;
        px[3] = -al * a / d1;  py[3] = al * c / d1;
        const uint32_t vote = (uint32_t)(m.len[i] * 65536.0f + 0.5f);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            if (!ok[s] || !ok[s + 1]) continue;
            const float x0 = roundf((px[s] + 1.f) * 0.5f * sc), y0 = roundf((py[s] + 1.f) * 0.5f * sc);
            const float x1 = roundf((px[s + 1] + 1.f) * 0.5f * sc), y1 = roundf((py[s + 1] + 1.f) * 0.5f * sc);
            if (!(x0 >= 0.f && x0 <= sc && y0 >= 0.f && y0 <= sc && x1 >= 0.f && x1 <= sc && y1 >= 0.f && y1 <= sc)) continue;
            const int steps = (int)roundf(fmaxf(fabsf(x1 - x0), fabsf(y1 - y0))) + 1;
            const float sx = steps > 1 ? (x1 - x0) / (float)(steps - 1) : 0.f;
            const float sy = steps > 1 ? (y1 - y0) / (float)(steps - 1) : 0.f;
            for (int j = lane; j < steps; j += 64) {
                const int xi = (int)roundf(x0 + (float)j * sx), yi = (int)roundf(y0 + (float)j * sy);
                atomicAdd(&s_acc[yi * d + xi], vote);
            }
            cast += (uint32_t)steps;
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < d * d; i += 256) {
        const uint32_t v = s_acc[i];
        if (v) atomicAdd(&acc[i], kSub ? (unsigned long long)0 - (unsigned long long)v : (unsigned long long)v);
    }
    if (n_votes && lane == 0 && cast) atomicAdd(n_votes, (unsigned long long)cast);
}

// argmax of the accumulator: the first maximum in row-major order (the oracle's cht_peak); out = {cell, value lo, value hi}
__global__ __launch_bounds__(1024) void cht_peak_kernel(const unsigned long long* __restrict__ acc, uint32_t cells,
                                                        uint32_t* __restrict__ out) {
    __shared__ unsigned long long s_v[16];
    __shared__ uint32_t s_i[16];
    unsigned long long bv = 0ull;
    uint32_t bi = 0xFFFFFFFFu;
    for (uint32_t i = threadIdx.x; i < cells; i += 1024) {
        const unsigned long long v = acc[i];
        if (bi == 0xFFFFFFFFu || v > bv) {  // (ascending i per thread: a later equal value does not replace)
            bv = v;
            bi = i;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const uint32_t olo = (uint32_t)__shfl_xor((int)(uint32_t)bv, off), ohi = (uint32_t)__shfl_xor((int)(uint32_t)(bv >> 32), off);
        const unsigned long long ov = ((unsigned long long)ohi << 32) | olo;
        const uint32_t oi = (uint32_t)__shfl_xor((int)bi, off);
        if (oi != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || ov > bv || (ov == bv && oi < bi))) {
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
            const unsigned long long ov = s_v[k];
            const uint32_t oi = s_i[k];
            if (oi != 0xFFFFFFFFu && (bi == 0xFFFFFFFFu || ov > bv || (ov == bv && oi < bi))) {
                bv = ov;
                bi = oi;
            }
        }
        out[0] = bi == 0xFFFFFFFFu ? 0u : bi;
        out[1] = (uint32_t)bv;
        out[2] = (uint32_t)(bv >> 32);
    }
}

}  // namespace

static int cht_check_size(int d) {
    if (d < 8 || d > kChtMax) {
        set_error("diamond-space accumulator: size must be in [8, 128]");
        return 1;
    }
    return 0;
}

int launch_cht_accumulate(PencilSoA m, uint32_t n, int d, unsigned long long* acc, hipStream_t s) {
    if (cht_check_size(d)) return 1;
    LR_HIP(hipMemsetAsync(acc, 0, (size_t)d * d * sizeof(unsigned long long), s));
    return launch_cht_votes(m, nullptr, n, d, acc, false, nullptr, s);
}

// adds (or takes back) the votes of n lines (all of the table if idx == nullptr) to an accumulator that exists
int launch_cht_votes(PencilSoA m, const uint32_t* idx, uint32_t n, int d, unsigned long long* acc, bool subtract,
                     unsigned long long* n_votes, hipStream_t s) {
    if (cht_check_size(d)) return 1;
    if (n == 0) return 0;
    const uint32_t blocks = (n + kChtLinesPerBlock - 1) / kChtLinesPerBlock;
    const size_t lds = (size_t)d * d * sizeof(uint32_t);
    if (subtract) hipLaunchKernelGGL(cht_votes_kernel<true>, dim3(blocks), dim3(256), lds, s, m, idx, n, d, acc, n_votes);
    else hipLaunchKernelGGL(cht_votes_kernel<false>, dim3(blocks), dim3(256), lds, s, m, idx, n, d, acc, n_votes);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_cht_peak(const unsigned long long* acc, int d, uint32_t* out3, hipStream_t s) {
    hipLaunchKernelGGL(cht_peak_kernel, dim3(1), dim3(1024), 0, s, acc, (uint32_t)(d * d), out3);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_refine_pairs(const void* seg, uint32_t n, void* edges, uint32_t* n_edges, uint32_t cap, hipStream_t s) {
    const uint32_t nb = (n + 63) / 64;
    hipLaunchKernelGGL(refine_pairs_kernel, dim3(nb, nb), dim3(256), 0, s, (const RefineSeg*)seg, n, (uint2*)edges,
                       n_edges, cap);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_prosac_count(PencilSoA m, uint32_t n, float tol, float degeneracy_tol, const uint32_t* sa,
                        const uint32_t* sb, uint32_t n_hyp, uint32_t* counts, hipStream_t s) {
    if (n_hyp == 0) return 0;
    if (n_hyp >= 16384u) {
        hipLaunchKernelGGL(prosac_count_kernel<8>, dim3((n_hyp + 31) / 32), dim3(256), 0, s, m, n, tol, degeneracy_tol, sa,
                           sb, n_hyp, counts);
    } else {
        hipLaunchKernelGGL(prosac_count_kernel<2>, dim3((n_hyp + 7) / 8), dim3(256), 0, s, m, n, tol, degeneracy_tol, sa, sb,
                           n_hyp, counts);
    }
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_prosac_records(PencilSoA m, uint32_t n, const uint32_t* sa, const uint32_t* sb, const uint32_t* counts,
                          uint32_t n_hyp, uint32_t best_in, float tol, uint32_t* rec, uint32_t cap, uint8_t* flags, hipStream_t s) {
    if (n_hyp == 0 || n == 0) return 0;
    hipLaunchKernelGGL(prosac_records_kernel, dim3(1), dim3(1024), 0, s, counts, n_hyp, best_in, rec, cap);
    hipLaunchKernelGGL(prosac_record_flags_kernel, dim3((n + 255) / 256, cap), dim3(256), 0, s, m, n, sa, sb, rec, cap, tol, flags);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_prosac_flags(PencilSoA m, uint32_t n, float px, float py, float pz, float tol, uint8_t* flags, hipStream_t s) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(prosac_flags_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, n, px, py, pz, tol, flags);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_ht_weights(PencilSoA m, uint32_t n, const int32_t* pa, const int32_t* pb, int n_pairs, int ht, float* peak,
                      float* weights, hipStream_t s) {
    if (ht > kHtMax || ht < 3) {
        set_error("launch_ht_weights: accumulator size out of range");
        return 1;
    }
    hipLaunchKernelGGL(ht_votes_kernel, dim3(1), dim3(1024), 0, s, m, pa, pb, n_pairs, ht, peak);
    hipLaunchKernelGGL(ht_weights_kernel, dim3((n + 255) / 256), dim3(256), 0, s, m, n, peak, weights);
    LR_HIP(hipGetLastError());
    return 0;
}

// Hypotheses per wavefront: eight when there are enough of them to fill the chip with waves anyway, else four
static void launch_score(PencilSoA m, uint32_t n, float tol, float degeneracy_tol, uint32_t n_iter, uint64_t seed,
                         uint32_t round, const uint32_t* gctl, int max_models, unsigned long long* best_slots, hipStream_t s) {
    // (sixteen per wavefront were measured too: 8.1e7 hypotheses/s against 1.1e8 with eight -- the registers of sixteen
    // accumulators and hypotheses halve the waves per SIMD)
    if (n_iter >= 65536u) {
        hipLaunchKernelGGL(ransac_score_kernel<8>, dim3((n_iter + 31) / 32), dim3(256), 0, s, m, n, tol, degeneracy_tol,
                           n_iter, seed, round, gctl, max_models, best_slots);
    } else {
        hipLaunchKernelGGL(ransac_score_kernel<4>, dim3((n_iter + 15) / 16), dim3(256), 0, s, m, n, tol, degeneracy_tol,
                           n_iter, seed, round, gctl, max_models, best_slots);
    }
}

// One solve: the scoring launch and a one-wavefront read-out; best score (bits) and iteration land in host_best[0..1]
// (page-locked memory).  best_slots: kRansacBestSlots zeroed 64-bit words, left zeroed again.
int launch_ransac_score(PencilSoA m, uint32_t n, float tol, float degeneracy_tol, uint32_t n_iter, uint64_t seed,
                        uint32_t round, unsigned long long* best_slots, uint32_t* host_best, hipStream_t s) {
    if (n < 2 || n_iter == 0) {
        set_error("launch_ransac_score: need at least 2 lines and 1 iteration");
        return 1;
    }
    launch_score(m, n, tol, degeneracy_tol, n_iter, seed, round, nullptr, 0, best_slots, s);
    hipLaunchKernelGGL(ransac_best_readout_kernel, dim3(1), dim3(64), 0, s, best_slots, host_best);
    LR_HIP(hipGetLastError());
    return 0;
}

// line count and round from the peeling control block (kernels_groups.hip)
int launch_ransac_score_dev(PencilSoA m, const uint32_t* gctl, int max_models, float tol, float degeneracy_tol,
                            uint32_t n_iter, uint64_t seed, unsigned long long* best_slots, hipStream_t s) {
    if (n_iter == 0) return 0;
    launch_score(m, 0u, tol, degeneracy_tol, n_iter, seed, 0u, gctl, max_models, best_slots, s);
    LR_HIP(hipGetLastError());
    return 0;
}


}  // namespace lramd
