// ORACLE — TEST INFRASTRUCTURE ONLY.
//
// CPU restatement of librectify's hot path (reference @ /root/reference, C++/Eigen/OpenMP),
// written from the source text because the reference cannot be built here (Eigen is an
// un-vendored, empty submodule: .gitmodules:1-3, src/CMakeLists.txt:5,29,36).
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library.  The product (librectify_amd/) never includes, links or calls anything here.
//
// Parity pins (tests/test_oracle_pins.py): doc/image.jpg_warp_lines.csv ->
// doc/image.jpg_warp_tform.csv (exact transform KAT), doc/image.jpg detector KAT at
// TRACE_TOLERANCE=0.3 (>=700/848 rows within 0.01 px), and the analytic KATs of
// src/test.cpp:19-39.  Everything else (group ids, refine, PROSAC, HT weights) is
// "parity unpinned": the reference holds no fixture for it (random_device seed,
// estimator.h:35) and this restatement is the definition.
//
// CANONICAL ARITHMETIC (the GPU path mirrors it bit for bit):
//  * fp32 wherever the reference is fp32; compiled with -ffp-contract=off, every fused
//    multiply-add is an explicit fmaf().
//  * 5x5 derivative-of-Gaussian correlation in its SEPARABLE form (the taps of filter.cpp:65-78 are a product d(x) g(y)):
//    a row pass and a column pass of explicit fmaf() chains, 16 operations a pixel -- conv_gradients below.  (The 25
//    separately rounded taps in row-major order are kept as conv_gradients_25tap, for the bound tests/test_oracle_pins.py puts
//    on the difference; the reference's own order under Eigen + -ffast-math is not defined.)
//  * every floating-point reduction over a pixel or line set is the "wave tree" T():
//    64 lane-strided sequential partial sums followed by an xor butterfly (32,16,..,1).
//  * peaks are ordered by (value desc, row asc, col asc): the reference uses an unstable
//    std::sort (filter.cpp:188) so ties are its one unspecified order.
//  * component pixels are put in row-major order before the line fit (the reference keeps
//    BFS order, which only changes float summation order).
//  * uninitialised grad_bin (line_detector.cpp:128) is defined as bin 0.
//  * RANSAC samples come from a counter-based generator (splitmix64 of seed, round,
//    iteration) instead of mt19937(random_device) + choice_knuth: same distribution
//    (uniform sorted pair), reproducible, and computable per hypothesis on the GPU.
//
// Every function cites the reference file:line it follows.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <map>
#include <queue>
#include <random>
#include <set>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace {

// ---------------------------------------------------------------------------------------
// config.h:7-59
constexpr float EPS = 1e-6f;
constexpr int MAX_MODELS = 4;
constexpr float ESTIMATOR_INLIER_MAX_ANGLE_DEG = 2.0f;
constexpr float ESTIMATOR_GARBAGE_MAX_ANGLE_DEG = 4.0f;
constexpr int RANSAC_MAX_ITER = 10000;
constexpr int EDGE_KERNEL_SIZE = 2;
constexpr float EDGE_KERNEL_SIGMA = 1.0f;
constexpr int SEED_DIST = 2;
constexpr float SEED_RATIO = 0.95f;  // config.h:44 is a double literal narrowed by the float parameter
constexpr float TRACE_TOLERANCE = 0.25f;
constexpr float LINE_MAX_ERR = 2.0f;
constexpr float LINE_MIN_LENGTH = 5.f;
constexpr int COMPONENT_MIN_SIZE = 5;

// librectify.h:44-54, :60-63, :79-86, :126-150
struct LineSegment {
    float x1, y1, x2, y2;
    float weight;
    float err;
    int group_id;
};
struct Point {
    float x, y, z;
};
struct ImageTransform {
    int width;
    int height;
    Point top_left, top_right, bottom_left, bottom_right;
    Point horizontal_vp;
    Point vertical_vp;
};
enum RectificationStrategy { ROTATE_H, ROTATE_V, RECTIFY, KEEP };
struct RectificationConfig {
    float vertical_vp_angular_tolerance;
    float vertical_vp_min_distance;
    int v_strategy;
    float horizontal_vp_min_distance;
    int h_strategy;
};

// threading.h:11-31
struct ThreadContext {
    int num_threads;
    explicit ThreadContext(int t) {
#ifdef _OPENMP
        num_threads = std::min(t, omp_get_max_threads());
#else
        num_threads = std::min(t, 1);
#endif
    }
    int get() const { return num_threads > 0 ? num_threads : 1; }
    bool enabled() const { return num_threads >= 0; }
};

struct V2 {
    float x, y;
};
struct V3 {
    float x, y, z;
};

// ---------------------------------------------------------------------------------------
// Canonical reduction: 64 lane-strided partial sums + xor butterfly.
template <class F>
float wave_tree_sum(size_t n, F term) {
    float lane[64];
    for (int j = 0; j < 64; ++j) {
        float acc = 0.0f;
        for (size_t i = j; i < n; i += 64) acc = acc + term(i);
        lane[j] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float nxt[64];
        for (int j = 0; j < 64; ++j) nxt[j] = lane[j] + lane[j ^ off];
        std::memcpy(lane, nxt, sizeof(lane));
    }
    return lane[0];
}

// ---------------------------------------------------------------------------------------
// filter.cpp:65-78
void gauss_deriv_kernel(int size, float sigma, bool dir_x, float* H) {
    int n = 2 * size + 1;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            float x = float(j - size);
            float y = float(i - size);
            float z = dir_x ? x : y;
            float a = float(2 * M_PI * std::pow(sigma, 4.0f));
            float e = std::exp(-(std::pow(x, 2.0f) + std::pow(y, 2.0f)) / (2 * std::pow(sigma, 2.0f)));
            H[i * n + j] = z / a * e;
        }
}

// filter.cpp:81-98 (conv_2d: correlation, zero border of kernel radius) -- as conv_gradients below.
// The taps of filter.cpp:65-78 are a product: H(i,j) = z/a * exp(-(x^2+y^2)/2s^2) = d(x) * g(y) for Hx (and
// d(y) * g(x) for Hy) with d(t) = t/a * exp(-t^2/2s^2), g(t) = exp(-t^2/2s^2).  The reference evaluates the 25-tap
// correlation with an Eigen block product whose summation order (under -ffast-math) is unspecified; the canonical
// arithmetic of this build is the separable form, which uses the (anti)symmetry of d and g:
//   row pass     hx = fma(I[x+2]-I[x-2], d2, (I[x+1]-I[x-1])*d1)      hs = fma(I[x+2]+I[x-2], g2, fma(I[x+1]+I[x-1], g1, I[x]))
//   column pass  dx = fma(hx[y+2]+hx[y-2], g2, fma(hx[y+1]+hx[y-1], g1, hx[y]))
//                dy = fma(hs[y+2]-hs[y-2], d2, (hs[y+1]-hs[y-1])*d1)
// (g(0) = exp(0) = 1, d(0) = 0, d(-t) = -d(t), g(-t) = g(t), all exactly in fp32.)  16 operations per pixel instead of
// 50; pinned by the detector KAT (tests/test_oracle_pins.py::test_pin2_*).
void gauss_deriv_factors(int size, float sigma, float* d, float* g) {
    int n = 2 * size + 1;
    for (int t = 0; t < n; ++t) {
        float x = float(t - size);
        float a = float(2 * M_PI * std::pow(sigma, 4.0f));
        float e = std::exp(-std::pow(x, 2.0f) / (2 * std::pow(sigma, 2.0f)));
        d[t] = x / a * e;
        g[t] = e;
    }
}

void conv_gradients(const float* img, int w, int h, float* dx, float* dy, const ThreadContext& ctx) {
    float d[5], g[5];
    gauss_deriv_factors(EDGE_KERNEL_SIZE, EDGE_KERNEL_SIGMA, d, g);
    const float d1 = d[3], d2 = d[4], g1 = g[3], g2 = g[4];
    // Frame-sized work planes are kept from call to call (per calling thread) and first touched inside the parallel loops
    // that fill them: allocating and zero-filling 33 MB vectors on the calling thread, call after call, cost the threaded
    // runs more than their loops took (round 3: "gradients" slower on 128 threads than on one).  Every element that is
    // read below is written first: rows 0 .. h-1, columns 2 .. w-3.
    static thread_local std::vector<float> tl_hx, tl_hs;  // (referred to through references: a worker's own would be empty)
    std::vector<float>&hx = tl_hx, &hs = tl_hs;
    if (hx.size() != size_t(w) * h) {
        hx = std::vector<float>();
        hs = std::vector<float>();
        hx.resize(size_t(w) * h);
        hs.resize(size_t(w) * h);
    }
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int r = 0; r < h; ++r) {
        std::fill(dx + size_t(r) * w, dx + size_t(r + 1) * w, 0.0f);  // conv_2d's zero border (filter.cpp:83)
        std::fill(dy + size_t(r) * w, dy + size_t(r + 1) * w, 0.0f);
        const float* I = img + size_t(r) * w;
        for (int x = 2; x < w - 2; ++x) {
            float a1 = I[x + 1] - I[x - 1], a2 = I[x + 2] - I[x - 2];
            float s1 = I[x + 1] + I[x - 1], s2 = I[x + 2] + I[x - 2];
            hx[size_t(r) * w + x] = std::fmaf(a2, d2, a1 * d1);
            hs[size_t(r) * w + x] = std::fmaf(s2, g2, std::fmaf(s1, g1, I[x]));
        }
    }
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x) {
            auto HX = [&](int k) { return hx[size_t(y + k) * w + x]; };
            auto HS = [&](int k) { return hs[size_t(y + k) * w + x]; };
            dx[size_t(y) * w + x] = std::fmaf(HX(2) + HX(-2), g2, std::fmaf(HX(1) + HX(-1), g1, HX(0)));
            dy[size_t(y) * w + x] = std::fmaf(HS(2) - HS(-2), d2, (HS(1) - HS(-1)) * d1);
        }
}

// The reference's own form (filter.cpp:81-98 with the taps of :65-78): out(i+2, j+2) = sum over the 5x5 block of
// block(a, b) * H(a, b), every tap H(a, b) = z / a * exp(-(x^2 + y^2) / 2 s^2) rounded on its own, summed here in
// row-major tap order with double accumulation (Eigen's order under -ffast-math is unspecified).  Not the canonical
// arithmetic of this build: kept as a second path so that tests/test_oracle_pins.py can bound what the separable form
// (which the product mirrors) may differ from it.
void conv_gradients_25tap(const float* img, int w, int h, float* dx, float* dy) {
    float Hx[25], Hy[25];
    gauss_deriv_kernel(EDGE_KERNEL_SIZE, EDGE_KERNEL_SIGMA, true, Hx);
    gauss_deriv_kernel(EDGE_KERNEL_SIZE, EDGE_KERNEL_SIGMA, false, Hy);
    std::fill(dx, dx + size_t(w) * h, 0.0f);
    std::fill(dy, dy + size_t(w) * h, 0.0f);
    for (int y = 2; y < h - 2; ++y)
        for (int x = 2; x < w - 2; ++x) {
            double sx = 0.0, sy = 0.0;
            for (int a = 0; a < 5; ++a)
                for (int b = 0; b < 5; ++b) {
                    const float v = img[size_t(y + a - 2) * w + (x + b - 2)];
                    sx += double(v * Hx[a * 5 + b]);
                    sy += double(v * Hy[a * 5 + b]);
                }
            dx[size_t(y) * w + x] = float(sx);
            dy[size_t(y) * w + x] = float(sy);
        }
}

// line_detector.cpp:41-49
void image_gradients(const float* img, int w, int h, float* dx, float* dy, float* mag, const ThreadContext& ctx) {
    conv_gradients(img, w, h, dx, dy, ctx);
    size_t n = size_t(w) * h;
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (long long i = 0; i < (long long)n; ++i) {
        float a = dx[i] * dx[i];
        float b = dy[i] * dy[i];
        mag[i] = std::sqrt(a + b);
    }
}

// line_detector.cpp:144-145: theta = float(i*M_PI)/n_bins; sin/cos of a float
void bin_trig(int n_bins, float* st, float* ct) {
    for (int i = 0; i < n_bins; ++i) {
        float theta = float(i * M_PI) / n_bins;
        st[i] = std::sin(theta);
        ct[i] = std::cos(theta);
    }
}

inline float directional(float dx, float dy, float s, float c) { return std::fabs(std::fmaf(dx, s, dy * c)); }

// filter.cpp:46-62
void binary_dilate(const uint8_t* in, int w, int h, uint8_t* out, const ThreadContext& ctx) {
    std::fill(out, out + size_t(w) * h, uint8_t(0));
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int i = 0; i < h - 2; ++i)
        for (int j = 0; j < w - 2; ++j) {
            uint8_t m = 0;
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) m |= in[size_t(i + a) * w + (j + b)];
            out[size_t(i + 1) * w + (j + 1)] = m != 0;
        }
}

// line_detector.cpp:126-182 — materialises the 8 planes exactly as the reference does.
void gradient_directions(const float* dx, const float* dy, int w, int h, int n_bins, int32_t* grad_bin,
                         std::vector<std::vector<float>>& grad, const ThreadContext& ctx) {
    size_t n = size_t(w) * h;
    static thread_local std::vector<float> tl_grad_max;  // (kept from call to call, as the planes are: see conv_gradients)
    std::vector<float>& grad_max = tl_grad_max;
    grad_max.assign(n, 0.0f);
    std::fill(grad_bin, grad_bin + n, 0);  // reference leaves it uninitialised (:128); canonical: 0
    if (int(grad.size()) != n_bins) grad.assign(n_bins, std::vector<float>());
    std::vector<float> st(n_bins), ct(n_bins);
    bin_trig(n_bins, st.data(), ct.data());
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int i = 0; i < n_bins; ++i) {
        if (grad[i].size() != n) {
            grad[i] = std::vector<float>();
            grad[i].resize(n);
        }
        for (size_t p = 0; p < n; ++p) grad[i][p] = directional(dx[p], dy[p], st[i], ct[i]);
    }
    // :152-156 serial in the reference
    for (int i = 0; i < n_bins; ++i) {
        const float* g = grad[i].data();
        for (size_t p = 0; p < n; ++p) {
            if (g[p] > grad_max[p]) grad_bin[p] = i;
            grad_max[p] = std::max(grad_max[p], g[p]);
        }
    }
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int i = 0; i < n_bins; ++i) {
        static thread_local std::vector<uint8_t> eq, mask;  // (per OpenMP thread)
        eq.resize(n);
        mask.resize(n);
        for (size_t p = 0; p < n; ++p) eq[p] = grad_bin[p] == i;
        binary_dilate(eq.data(), w, h, mask.data(), ThreadContext(-1));
        for (size_t p = 0; p < n; ++p)
            if (!mask[p]) grad[i][p] = 0.0f;
    }
}

// filter.cpp:29-43
void maximum_filter(const float* img, int w, int h, int size, float* out, const ThreadContext& ctx) {
    int n = 2 * size + 1;
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int i = 0; i < h; ++i) std::fill(out + size_t(i) * w, out + size_t(i + 1) * w, 0.0f);
#pragma omp parallel for num_threads(ctx.get()) if (ctx.enabled())
    for (int i = 0; i < h - n + 1; ++i)
        for (int j = 0; j < w - n + 1; ++j) {
            float m = img[size_t(i) * w + j];
            for (int a = 0; a < n; ++a)
                for (int b = 0; b < n; ++b) m = std::max(m, img[size_t(i + a) * w + (j + b)]);
            out[size_t(i + size) * w + (j + size)] = m;
        }
}

struct PeakPoint {
    int i, j;
    float v;
};

// filter.cpp:161-195; tie order made canonical (value desc, row asc, col asc)
std::vector<PeakPoint> find_peaks(const float* img, int w, int h, int size, float min_value, const ThreadContext& ctx) {
    static thread_local std::vector<float> tl_max_im;
    std::vector<float>& max_im = tl_max_im;
    max_im.resize(size_t(w) * h);
    maximum_filter(img, w, h, size, max_im.data(), ctx);
    std::vector<PeakPoint> res;
    res.reserve(1024);
    for (int i = 0; i < h; ++i)
        for (int j = 0; j < w; ++j) {
            size_t p = size_t(i) * w + j;
            if (max_im[p] == img[p] && img[p] > min_value) res.push_back({i, j, img[p]});
        }
    std::stable_sort(res.begin(), res.end(), [](const PeakPoint& a, const PeakPoint& b) { return a.v > b.v; });
    return res;
}

struct Component {
    std::vector<int32_t> px;  // linear indices, BFS order
};

// filter.cpp:101-153 + line_detector.cpp:92-122.  `label` (optional) receives, per pixel, the
// index of the seed whose flood claimed it (small floods included), else -1.
std::vector<Component> find_components(const std::vector<std::vector<float>>& grad, int w, int h,
                                       const std::vector<PeakPoint>& seed, const std::vector<int>& seed_bin,
                                       float tolerance, int32_t* label, std::vector<int>* comp_seed) {
    std::vector<uint8_t> visited(size_t(w) * h, 0);
    for (int j = 0; j < w; ++j) visited[j] = visited[size_t(h - 1) * w + j] = 1;
    for (int i = 0; i < h; ++i) visited[size_t(i) * w] = visited[size_t(i) * w + w - 1] = 1;
    if (label) std::fill(label, label + size_t(w) * h, -1);
    std::vector<Component> components;
    static const int dr[8] = {0, 0, 1, -1, -1, -1, 1, 1};
    static const int dc[8] = {-1, 1, 0, 0, -1, 1, -1, 1};
    for (size_t k = 0; k < seed.size(); ++k) {
        int sr = seed[k].i, sc = seed[k].j;
        if (visited[size_t(sr) * w + sc]) continue;
        const float* image = grad[seed_bin[k]].data();
        float seed_val = image[size_t(sr) * w + sc];
        float min_val = (1 - tolerance) * seed_val;
        Component c;
        std::queue<std::pair<int, int>> q;
        q.push({sr, sc});
        while (!q.empty()) {
            auto p = q.front();
            q.pop();
            int r = p.first, cc = p.second;
            size_t li = size_t(r) * w + cc;
            if (!visited[li] && image[li] > min_val) {
                c.px.push_back(int32_t(li));
                visited[li] = 1;
                if (label) label[li] = int32_t(k);
                for (int d = 0; d < 8; ++d) q.push({r + dr[d], cc + dc[d]});
            }
        }
        if (int(c.px.size()) > COMPONENT_MIN_SIZE) {
            components.emplace_back(std::move(c));
            if (comp_seed) comp_seed->push_back(int(k));
        }
    }
    return components;
}

// Symmetric 2x2 eigen-decomposition (replaces Eigen::SelfAdjointEigenSolver<Matrix2f>,
// geometry.cpp:37).  Closed form in double, sqrt/div only.  Returns the unit major
// eigenvector (row, col) with the sign rule observed on all 848 golden rows: the p1->p2
// direction has row-component >= col-component.
inline void major_axis_2x2(float a_, float b_, float c_, float& d_r, float& d_c) {
    double a = a_, b = b_, c = c_;
    double hd = (a - c) * 0.5;
    double rad = std::sqrt(hd * hd + b * b);
    double lmax = (a + c) * 0.5 + rad;
    double vr, vc;
    if (a >= c) {
        vr = lmax - c;
        vc = b;
    } else {
        vr = b;
        vc = lmax - a;
    }
    double nn = std::sqrt(vr * vr + vc * vc);
    if (nn > 0.0) {
        vr = vr / nn;
        vc = vc / nn;
    } else {
        vr = 1.0;
        vc = 0.0;
    }
    float fr = float(vr), fc = float(vc);
    if (fr < fc || (fr == fc && fr < 0.0f)) {
        fr = -fr;
        fc = -fc;
    }
    d_r = fr;
    d_c = fc;
}

// geometry.cpp:20-61.  X = (row, col), w = pixel values.
LineSegment fit_line_parameters(const float* Xr, const float* Xc, const float* w, size_t n) {
    float S = wave_tree_sum(n, [&](size_t i) { return w[i]; });
    std::vector<float> wn(n);
    for (size_t i = 0; i < n; ++i) wn[i] = w[i] / S;
    float a_r = wave_tree_sum(n, [&](size_t i) { return wn[i] * Xr[i]; });
    float a_c = wave_tree_sum(n, [&](size_t i) { return wn[i] * Xc[i]; });
    std::vector<float> cr(n), cc(n);
    for (size_t i = 0; i < n; ++i) {
        cr[i] = Xr[i] - a_r;
        cc[i] = Xc[i] - a_c;
    }
    float cov_rr = wave_tree_sum(n, [&](size_t i) { return (cr[i] * wn[i]) * cr[i]; });
    float cov_rc = wave_tree_sum(n, [&](size_t i) { return (cr[i] * wn[i]) * cc[i]; });
    float cov_cc = wave_tree_sum(n, [&](size_t i) { return (cc[i] * wn[i]) * cc[i]; });
    float d_r, d_c;
    major_axis_2x2(cov_rr, cov_rc, cov_cc, d_r, d_c);
    float n_r = -d_c, n_c = d_r;
    float t0 = INFINITY, t1 = -INFINITY;
    for (size_t i = 0; i < n; ++i) {
        float t = cr[i] * d_r + cc[i] * d_c;
        t0 = std::min(t0, t);
        t1 = std::max(t1, t);
    }
    float esum = wave_tree_sum(n, [&](size_t i) { return std::fabs(cr[i] * n_r + cc[i] * n_c); });
    LineSegment l;
    l.x1 = a_c + d_c * t0;
    l.y1 = a_r + d_r * t0;
    l.x2 = a_c + d_c * t1;
    l.y2 = a_r + d_r * t1;
    l.weight = S / float(n);
    l.err = esum / float(n);
    l.group_id = -1;
    return l;
}

struct StageTimes {
    double gradients, directions, seeds, components, fitting;
};

// line_detector.cpp:185-251
std::vector<LineSegment> find_line_segments(const float* img, int w, int h, int seed_dist, float seed_ratio,
                                            float mag_tolerance, const ThreadContext& ctx, int32_t* label_out,
                                            std::vector<PeakPoint>* seeds_out, std::vector<int>* comp_seed_out,
                                            StageTimes* times) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    size_t n = size_t(w) * h;
    // (work planes kept from call to call: see conv_gradients.  The names below are references to the CALLING thread's
    // planes: inside an OpenMP region a thread_local name would mean the worker's own, empty ones.)
    static thread_local std::vector<float> tl_dx, tl_dy, tl_mag;
    static thread_local std::vector<int32_t> tl_grad_bin;
    static thread_local std::vector<std::vector<float>> tl_grad;
    std::vector<float>&dx = tl_dx, &dy = tl_dy, &mag = tl_mag;
    std::vector<int32_t>& grad_bin = tl_grad_bin;
    std::vector<std::vector<float>>& grad = tl_grad;
    if (dx.size() != n) {
        for (auto* v : {&dx, &dy, &mag}) {
            *v = std::vector<float>();
            v->resize(n);
        }
        grad_bin.resize(n);
    }
    image_gradients(img, w, h, dx.data(), dy.data(), mag.data(), ctx);
    auto t1 = clk::now();
    gradient_directions(dx.data(), dy.data(), w, h, 8, grad_bin.data(), grad, ctx);
    auto t2 = clk::now();
    float mx = 0.0f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, mag[i]);
    float min_seed_value = mx * (1 - std::max(std::min(seed_ratio, 1.f), 0.f));
    auto seed = find_peaks(mag.data(), w, h, seed_dist, min_seed_value, ctx);
    std::vector<int> seed_bin(seed.size());
    for (size_t i = 0; i < seed.size(); ++i) seed_bin[i] = grad_bin[size_t(seed[i].i) * w + seed[i].j];
    auto t3 = clk::now();
    std::vector<int> comp_seed;
    auto components = find_components(grad, w, h, seed, seed_bin, mag_tolerance, label_out, &comp_seed);
    auto t4 = clk::now();
    // line_detector.cpp:66-89 (serial order = seed order)
    std::vector<LineSegment> res(components.size());
#pragma omp parallel for schedule(dynamic, 16) num_threads(ctx.get()) if (ctx.enabled())
    for (long long ci = 0; ci < (long long)components.size(); ++ci) {
        auto px = components[ci].px;
        std::sort(px.begin(), px.end());  // canonical order: row-major
        const float* image = grad[seed_bin[comp_seed[ci]]].data();
        std::vector<float> xr(px.size()), xc(px.size()), pv(px.size());
        for (size_t i = 0; i < px.size(); ++i) {
            xr[i] = float(px[i] / w);
            xc[i] = float(px[i] % w);
            pv[i] = image[px[i]];
        }
        res[ci] = fit_line_parameters(xr.data(), xc.data(), pv.data(), px.size());
    }
    auto t5 = clk::now();
    if (times) {
        auto ms = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        times->gradients = ms(t0, t1);
        times->directions = ms(t1, t2);
        times->seeds = ms(t2, t3);
        times->components = ms(t3, t4);
        times->fitting = ms(t4, t5);
    }
    if (seeds_out) *seeds_out = seed;
    if (comp_seed_out) *comp_seed_out = comp_seed;
    return res;
}

// ---------------------------------------------------------------------------------------
// geometry.cpp helpers
inline float seg_length(const LineSegment& l) {  // :90-93
    float dx = l.x2 - l.x1, dy = l.y2 - l.y1;
    return std::sqrt(dx * dx + dy * dy);
}

struct BBox {
    float minx, miny, maxx, maxy;
};
BBox bounding_box(const std::vector<LineSegment>& lines) {  // :96-112
    BBox b{INFINITY, INFINITY, -INFINITY, -INFINITY};
    for (auto& l : lines) {
        b.minx = std::min(b.minx, std::min(l.x1, l.x2));
        b.miny = std::min(b.miny, std::min(l.y1, l.y2));
        b.maxx = std::max(b.maxx, std::max(l.x1, l.x2));
        b.maxy = std::max(b.maxy, std::max(l.y1, l.y2));
    }
    return b;
}
inline V2 bbox_size(const BBox& b) { return {b.maxx - b.minx, b.maxy - b.miny}; }  // :279-282
inline V2 bbox_center(const BBox& b) {                                             // :272-276
    V2 s = bbox_size(b);
    return {b.minx + 0.5f * s.x, b.miny + 0.5f * s.y};
}
std::vector<LineSegment> normalize_lines(const std::vector<LineSegment>& lines, V2 p, float s) {  // :258-270
    std::vector<LineSegment> out(lines);
    for (auto& l : out) {
        l.x1 = (l.x1 - p.x) / s;
        l.y1 = (l.y1 - p.y) / s;
        l.x2 = (l.x2 - p.x) / s;
        l.y2 = (l.y2 - p.y) / s;
    }
    return out;
}
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalized3(V3 v) {  // Eigen MatrixBase::normalized(): guarded by z > 0
    float z = (v.x * v.x + v.y * v.y) + v.z * v.z;
    if (z > 0.0f) {
        float n = std::sqrt(z);
        return {v.x / n, v.y / n, v.z / n};
    }
    return v;
}
inline V2 normalized2(V2 v) {
    float z = v.x * v.x + v.y * v.y;
    if (z > 0.0f) {
        float n = std::sqrt(z);
        return {v.x / n, v.y / n};
    }
    return v;
}
inline V3 normalize_point(V3 p) {  // :232-238
    if (std::fabs(p.z) < EPS) return {p.x, p.y, 0.f};
    return {p.x / p.z, p.y / p.z, 1.f};
}
inline V2 direction(V3 a, V3 b) { return normalized2({a.x - b.x * a.z, a.y - b.y * a.z}); }  // :240-245
inline float distance(V3 a, V3 b) {                                                           // :247-255
    if (a.z < EPS || b.z < EPS) return INFINITY;
    float dx = a.x - b.x, dy = a.y - b.y;
    return std::sqrt(dx * dx + dy * dy);
}

// geometry.cpp:214-229 for one (anchor, direction) row; rowwise().normalize() has no zero guard.
inline float inclination1(float ax, float ay, float dx, float dy, V3 p) {
    float vx, vy;
    if (std::fabs(p.z) < EPS) {
        vx = p.x;
        vy = p.y;
    } else {
        float pnx = p.x / p.z, pny = p.y / p.z;
        vx = pnx - ax;
        vy = pny - ay;
    }
    float nn = vx * vx + vy * vy;
    float nrm = std::sqrt(nn);
    float ux = vx / nrm, uy = vy / nrm;
    return std::fabs(ux * dx + uy * dy);
}

// Symmetric 3x3 eigen-decomposition by cyclic Jacobi in double (replaces
// SelfAdjointEigenSolver<Matrix3f>, line_pencil.cpp:123).  Returns the eigenvector of the
// smallest eigenvalue, cast to float.
V3 min_eigvec_3x3(const float cov[9]) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) A[i][j] = 0.5 * (double(cov[i * 3 + j]) + double(cov[j * 3 + i]));
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = A[0][1] * A[0][1] + A[0][2] * A[0][2] + A[1][2] * A[1][2];
        // converged: the off-diagonal part is below double precision relative to the diagonal (waiting for it to
        // underflow to exactly zero can take all 64 sweeps and changes nothing in the float result)
        double dg = A[0][0] * A[0][0] + A[1][1] * A[1][1] + A[2][2] * A[2][2];
        if (off <= 1e-36 * dg) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (A[p][q] == 0.0) continue;
                double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq;
                    V[k][q] = s * vkp + c * vkq;
                }
            }
    }
    int k = 0;
    if (A[1][1] < A[k][k]) k = 1;
    if (A[2][2] < A[k][k]) k = 2;
    double nx = V[0][k], ny = V[1][k], nz = V[2][k];
    double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
    return {float(nx / nn), float(ny / nn), float(nz / nn)};
}

// ---------------------------------------------------------------------------------------
// line_pencil.{h,cpp}
struct LinePencilModel {
    std::vector<V3> h;
    std::vector<V2> anchor, direction;
    std::vector<float> length;
    float degeneracy_tol = 0.05f;
    int ht_space_size = 65;
    int ht_num_hypotheses = 20000;

    explicit LinePencilModel(const std::vector<LineSegment>& lines) {  // :25-32
        size_t n = lines.size();
        h.resize(n);
        anchor.resize(n);
        direction.resize(n);
        length.resize(n);
        for (size_t i = 0; i < n; ++i) {
            const auto& l = lines[i];
            h[i] = normalized3(cross({l.x1, l.y1, 1.f}, {l.x2, l.y2, 1.f}));  // geometry.cpp:64-69
            anchor[i] = {(l.x2 + l.x1) / 2, (l.y2 + l.y1) / 2};              // :72-75
            float dx = l.x2 - l.x1, dy = l.y2 - l.y1;                         // :78-81
            float len = std::sqrt(dx * dx + dy * dy);
            length[i] = len;
            direction[i] = {dx / len, dy / len};  // rowwise().normalize(): no zero guard
        }
    }
    int size() const { return int(h.size()); }
    bool sample_check(int a, int b) const {  // :89-98
        float dx = h[a].x - h[b].x, dy = h[a].y - h[b].y, dz = h[a].z - h[b].z;
        return std::sqrt((dx * dx + dy * dy) + dz * dz) > degeneracy_tol;
    }
    V3 fit(int a, int b) const { return cross(h[a], h[b]); }  // :101-108
    V3 fit_optimal(const std::vector<int>& idx) const {       // :111-128 (empty set => all lines)
        // cov = h^T diag(length) h over the index set; each of the nine sums is the canonical tree (the reference's
        // Eigen product leaves the order unspecified)
        const size_t m = idx.empty() ? size_t(size()) : idx.size();
        float cov[9];
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b)
                cov[a * 3 + b] = wave_tree_sum(m, [&](size_t j) {
                    int i = idx.empty() ? int(j) : idx[j];
                    float hv[3] = {h[i].x, h[i].y, h[i].z};
                    float t = hv[a] * length[i];
                    return t * hv[b];
                });
        return min_eigvec_3x3(cov);
    }
    float error1(V3 hyp, int i) const {  // :131-134
        return -inclination1(anchor[i].x, anchor[i].y, direction[i].x, direction[i].y, hyp) + 1.0f;
    }
    float inlier_score(V3 hyp, float tol, const std::vector<int>& idx) const {  // :137-140
        return wave_tree_sum(idx.size(), [&](size_t j) {
            int i = idx[j];
            return (error1(hyp, i) < tol) ? length[i] : 0.0f;
        });
    }
    // :47-86 — Hough weights on the unit hemisphere; mt19937 default seed (5489)
    std::vector<float> get_weights(const std::vector<int>& idx) const {
        float k = std::floor(ht_space_size / 2.f);
        float k1 = k - 1;
        std::vector<float> acc(size_t(ht_space_size) * ht_space_size, 0.f);
        std::mt19937 rng;
        std::uniform_int_distribution<int> rand_idx(0, int(idx.size()) - 1);
        for (int i = 0; i < ht_num_hypotheses; ++i) {
            int a = idx[rand_idx(rng)];
            int b = idx[rand_idx(rng)];
            V3 x = cross(h[a], h[b]);
            if (std::fabs(x.x) < 0.0001f && std::fabs(x.y) < 0.0001f && std::fabs(x.z) < 0.0001f) continue;
            x = normalized3(x);
            if (x.z < 0.f) x = {-x.x, -x.y, -x.z};
            int u = int(std::round(k1 * x.x + k));
            int v = int(std::round(k1 * x.y + k));
            acc[size_t(u) * ht_space_size + v] += length[a] + length[b];
        }
        int max_u = 0, max_v = 0;
        float best = acc[0];
        // Eigen's maxCoeff visitor walks a column-major ArrayXXf column by column, first strict max wins
        for (int v = 0; v < ht_space_size; ++v)
            for (int u = 0; u < ht_space_size; ++u)
                if (acc[size_t(u) * ht_space_size + v] > best) {
                    best = acc[size_t(u) * ht_space_size + v];
                    max_u = u;
                    max_v = v;
                }
        V3 p{(max_u - k) / k1, (max_v - k) / k1, 0.f};
        float pn = std::sqrt((p.x * p.x + p.y * p.y) + p.z * p.z);
        if (pn > 1.f) p = {p.x / pn, p.y / pn, p.z / pn};
        p.z = std::sqrt(1.f - (std::pow(p.x, 2.f) + std::pow(p.y, 2.f)));
        std::vector<float> wts(idx.size());
        for (size_t j = 0; j < idx.size(); ++j) {
            int i = idx[j];
            float inc = inclination1(anchor[i].x, anchor[i].y, direction[i].x, direction[i].y, p);
            wts[j] = std::pow(inc, 4.0f);
        }
        return wts;
    }
};

inline float cos_threshold(float deg) { return 1.0f - std::cos(deg / 180.f * float(M_PI)); }  // :143-146

// Counter-based sample generator (canonical replacement of choice_knuth under
// mt19937(random_device), estimator.h:35,49-50): uniform sorted pair out of n.
inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
inline void sample_pair(uint64_t seed, uint32_t round, uint32_t iter, uint32_t n, uint32_t& a, uint32_t& b) {
    uint64_t z = splitmix64(seed ^ splitmix64((uint64_t(round) << 32) | iter));
    uint32_t u1 = uint32_t(z), u2 = uint32_t(z >> 32);
    uint32_t i = uint32_t((uint64_t(u1) * n) >> 32);
    uint32_t j = uint32_t((uint64_t(u2) * (n - 1)) >> 32);
    if (j >= i) ++j;
    a = std::min(i, j);
    b = std::max(i, j);
}

// estimator.h:37-78 with the canonical sampler; serial "first strictly better wins".
V3 ransac_solve(const LinePencilModel& model, const std::vector<int>& indices, float tol, int n_iter, uint64_t seed,
                uint32_t round, const ThreadContext& ctx, V3* best_raw = nullptr, float* best_score_out = nullptr,
                int* best_iter_out = nullptr) {
    V3 best_h{0.f, 0.f, 0.f};
    float best = 0.f;
    int best_iter = -1;
    int n = int(indices.size());
    if (n >= 2) {
        std::vector<float> score(n_iter, -1.0f);
        std::vector<V3> hyp(n_iter);
#pragma omp parallel for schedule(dynamic, 64) num_threads(ctx.get()) if (ctx.enabled())
        for (int i = 0; i < n_iter; ++i) {
            uint32_t a, b;
            sample_pair(seed, round, uint32_t(i), uint32_t(n), a, b);
            int ia = indices[a], ib = indices[b];
            if (!model.sample_check(ia, ib)) continue;
            V3 h = model.fit(ia, ib);
            hyp[i] = h;
            score[i] = model.inlier_score(h, tol, indices);
        }
        for (int i = 0; i < n_iter; ++i)
            if (score[i] > best) {
                best = score[i];
                best_h = hyp[i];
                best_iter = i;
            }
    }
    if (best_raw) *best_raw = best_h;
    if (best_score_out) *best_score_out = best;
    if (best_iter_out) *best_iter_out = best_iter;
    std::vector<int> inl;
    for (int i : indices)
        if (model.error1(best_h, i) < tol) inl.push_back(i);
    return model.fit_optimal(inl);
}

// ---------------------------------------------------------------------------------------
// prosac.h — PROSAC_Estimator (compiled header in the reference, never instantiated: ChangeLog.md:1-2;
// parity unpinned, this restatement is the definition).
//
// Canonical deviations, all documented in DESIGN.md:
//  * samples come from the counter-based generator (as for RANSAC);
//  * prosac.h:186 writes sample(m-1) = n, one past U_n (and out of range when n == N); the
//    intended index n-1 is used;
//  * the Hough votes of get_weights are accumulated in 2^-20 fixed point (order independent),
//    the vote pairs still come from std::mt19937 default-seeded as in line_pencil.cpp:53-59.
inline int niter_RANSAC(double p, double epsilon, int s, int Nmax) {  // prosac.h:31-55
    if (Nmax == -1) Nmax = INT32_MAX;
    if (epsilon <= 0.) return 1;
    double logarg = -std::exp(s * std::log(1. - epsilon));
    double logval = std::log(1. + logarg);
    double N = std::log(1. - p) / logval;
    if (logval < 0. && N < Nmax) return (int)std::ceil(N);
    return Nmax;
}

static const float chi2_table[20] = {INFINITY,   6.6348966f,  5.41189443f, 4.70929225f, 4.21788459f, 3.84145882f, 3.5373846f,
                                     3.28302029f, 3.06490172f, 2.8743734f,  2.70554345f, 2.55422131f, 2.41732093f, 2.29250453f,
                                     2.17795916f, 2.07225086f, 1.97422609f, 1.88294329f, 1.79762406f, 1.71761761f};

struct ProsacParams {
    float eta = 0.05f, beta = 0.01f, psi = 0.02f, p_good_sample = 0.9f, max_outlier_proportion = 0.5f;
    int T_N = -1;  // <= 0: niter_RANSAC(p_good_sample, max_outlier_proportion, m, -1) as prosac.h:116 (= 9)
};

// fixed-point Hough accumulator variant of LinePencilModel::get_weights (line_pencil.cpp:47-86)
std::vector<float> get_weights_fixed(const LinePencilModel& M, const std::vector<int>& idx) {
    const int S = M.ht_space_size;
    float k = std::floor(S / 2.f), k1 = k - 1;
    std::vector<uint64_t> acc(size_t(S) * S, 0);
    std::mt19937 rng;
    std::uniform_int_distribution<int> rand_idx(0, int(idx.size()) - 1);
    for (int i = 0; i < M.ht_num_hypotheses; ++i) {
        int a = idx[rand_idx(rng)];
        int b = idx[rand_idx(rng)];
        V3 x = cross(M.h[a], M.h[b]);
        if (std::fabs(x.x) < 0.0001f && std::fabs(x.y) < 0.0001f && std::fabs(x.z) < 0.0001f) continue;
        x = normalized3(x);
        if (x.z < 0.f) x = {-x.x, -x.y, -x.z};
        int u = int(std::round(k1 * x.x + k));
        int v = int(std::round(k1 * x.y + k));
        float vote = M.length[a] + M.length[b];
        acc[size_t(u) * S + v] += uint64_t(vote * 1048576.0f + 0.5f);
    }
    int max_u = 0, max_v = 0;
    uint64_t best = acc[0];
    for (int v = 0; v < S; ++v)  // column-major first strict max, as Eigen's maxCoeff visitor
        for (int u = 0; u < S; ++u)
            if (acc[size_t(u) * S + v] > best) {
                best = acc[size_t(u) * S + v];
                max_u = u;
                max_v = v;
            }
    V3 p{(max_u - k) / k1, (max_v - k) / k1, 0.f};
    float pn = std::sqrt((p.x * p.x + p.y * p.y) + p.z * p.z);
    if (pn > 1.f) p = {p.x / pn, p.y / pn, p.z / pn};
    p.z = std::sqrt(1.f - (p.x * p.x + p.y * p.y));
    std::vector<float> wts(idx.size());
    for (size_t j = 0; j < idx.size(); ++j) {
        int i = idx[j];
        float inc = inclination1(M.anchor[i].x, M.anchor[i].y, M.direction[i].x, M.direction[i].y, p);
        float i2 = inc * inc;
        wts[j] = i2 * i2;  // pow(4)
    }
    return wts;
}

// one uniform index out of n, counter based (second word of the pair generator's hash is unused)
inline uint32_t sample_one(uint64_t seed, uint32_t round, uint32_t iter, uint32_t n) {
    uint64_t z = splitmix64(seed ^ splitmix64((uint64_t(round) << 32) | iter));
    return uint32_t((uint64_t(uint32_t(z)) * n) >> 32);
}

struct ProsacTrace {
    int iterations = 0, n_star = 0, best_iter = -1, I_N_best = 0;
};

// prosac.h:104-299
V3 prosac_solve(const LinePencilModel& model, const std::vector<int>& indices, float tol, const ProsacParams& P,
                uint64_t seed, uint32_t round, ProsacTrace* trace = nullptr) {
    std::vector<float> weights = get_weights_fixed(model, indices);
    std::vector<int> order(indices.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = int(i);
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return weights[a] > weights[b]; });  // utils.h:36-44
    std::vector<int> idx(indices.size());
    for (size_t i = 0; i < order.size(); ++i) idx[i] = indices[order[i]];
    const int N = int(idx.size());
    const int m = 2;
    const int T_N = P.T_N > 0 ? P.T_N : niter_RANSAC(P.p_good_sample, P.max_outlier_proportion, m, -1);
    float chi2_value;
    {
        float p2 = 2 * P.psi;
        int ci = int(std::floor(std::max(std::min(p2, 0.2f), 0.01f) * 100));
        chi2_value = chi2_table[ci];
    }
    auto Imin = [&](int mm, int n) {
        double mu = n * P.beta;
        double sigma = std::sqrt(n * P.beta * (1 - P.beta));
        return (int)std::ceil(mm + mu + sigma * std::sqrt(chi2_value));
    };
    int n_star = N, I_n_star = 0, I_N_best = 0, t = 0, n = m, T_n_prime = 1, k_n_star = T_N;
    const int I_N_min = int((1. - P.max_outlier_proportion) * N);
    double T_n = T_N;
    for (int i = 0; i < m; i++) T_n *= (double)(n - i) / (N - i);
    V3 p_best{0, 0, 0};
    std::vector<uint8_t> best_inl(N, 0);
    int best_iter = -1;
    std::vector<uint8_t> isInlier(N);
    while (((I_N_best < I_N_min) || t <= k_n_star) && t < T_N) {
        t = t + 1;
        if ((t > T_n_prime) && (n < n_star)) {
            double T_nplus1 = (T_n * (n + 1)) / (n + 1 - m);
            n = n + 1;
            T_n_prime = T_n_prime + (int)std::ceil(T_nplus1 - T_n);
            T_n = T_nplus1;
        }
        int sa, sb;
        if (t > T_n_prime) {
            uint32_t a, b;
            sample_pair(seed, round, uint32_t(t), uint32_t(n), a, b);
            sa = int(a);
            sb = int(b);
        } else {
            sa = int(sample_one(seed, round, uint32_t(t), uint32_t(n - 1)));
            sb = n - 1;
        }
        int ia = idx[sa], ib = idx[sb];
        if (!model.sample_check(ia, ib)) continue;
        V3 p_t = model.fit(ia, ib);
        int I_N = 0;
        for (int i = 0; i < N; ++i) {
            isInlier[i] = model.error1(p_t, idx[i]) < tol;
            I_N += isInlier[i];
        }
        if (I_N > I_N_best) {
            I_N_best = I_N;
            p_best = p_t;
            best_inl = isInlier;
            best_iter = t;
            int n_best = N, I_n_best = I_N;
            double epsilon_n_best = (double)I_n_best / n_best;
            int n_test, I_n_test;
            for (n_test = N, I_n_test = I_N; n_test > m; n_test--) {
                if ((I_n_test * n_best > I_n_best * n_test) &&
                    (I_n_test > epsilon_n_best * n_test + std::sqrt(n_test * epsilon_n_best * (1. - epsilon_n_best) * 2.706))) {
                    if (I_n_test < Imin(m, n_test)) break;
                    n_best = n_test;
                    I_n_best = I_n_test;
                    epsilon_n_best = (double)I_n_best / n_best;
                }
                I_n_test -= isInlier[n_test - 1];
            }
            if (I_n_best * n_star > I_n_star * n_best) {
                n_star = n_best;
                I_n_star = I_n_best;
                k_n_star = niter_RANSAC(1. - P.eta, 1. - I_n_star / (double)n_star, m, T_N);
            }
        }
    }
    if (trace) {
        trace->iterations = t;
        trace->n_star = n_star;
        trace->best_iter = best_iter;
        trace->I_N_best = I_N_best;
    }
    std::vector<int> inl;
    for (int i = 0; i < N; ++i)
        if (best_inl[i]) inl.push_back(idx[i]);
    return model.fit_optimal(inl);
}

// estimator.h:99-145
std::vector<int> estimate_multiple_structures(const LinePencilModel& model, int max_structures, float tol,
                                              float garbage_tol, int n_iter, uint64_t seed, const ThreadContext& ctx,
                                              std::vector<V3>* models_out = nullptr) {
    int N = model.size();
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int num_observations = N;
    int k = 0;
    while (num_observations >= 2 && k < max_structures) {
        std::vector<int> obs;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0) obs.push_back(i);
        V3 h = ransac_solve(model, obs, tol, n_iter, seed, uint32_t(k), ctx);
        if (models_out) models_out->push_back(h);
        int n_in = 0, n_gb = 0;
        for (int i : obs) {
            float e = model.error1(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
            }
        }
        num_observations -= n_in + n_gb;
        ++k;
    }
    for (int i = 0; i < N; ++i)
        if (garbage_flag[i] == 1) inlier_flag[i] = -1;
    return inlier_flag;
}

// line_pencil.cpp:148-177
void estimate_line_pencils(std::vector<LineSegment>& lines, int max_models, float inlier_deg, float garbage_deg,
                           int n_iter, uint64_t seed, const ThreadContext& ctx, std::vector<V3>* models_out = nullptr) {
    BBox bb = bounding_box(lines);
    V2 p = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    auto lines_norm = normalize_lines(lines, p, scale);
    float inlier_tol = cos_threshold(inlier_deg);
    float garbage_tol = cos_threshold(garbage_deg);
    LinePencilModel model(lines_norm);
    auto groups = estimate_multiple_structures(model, max_models, inlier_tol, garbage_tol, n_iter, seed, ctx, models_out);
    for (size_t i = 0; i < lines.size(); ++i) lines[i].group_id = groups[i];
}

// ---------------------------------------------------------------------------------------
// cht.{h,cpp} — diamond-space ("cascaded Hough") vanishing point accumulator.  The reference's version is an
// uncompiled sketch (insert() uses undeclared names, argmax() is empty: SURVEY.md §0.1), so this follows what
// cht.h:13-24 says it should do, with the published diamond-space mapping (Dubska & Herout 2013):
//   line (a,b,c) -> polyline through  [alpha*a/(c+gamma*a), -alpha*c/(c+gamma*a)]  [b/(c+beta*b), 0]
//                   [0, b/(a+alpha*b)]  [-alpha*a/(c+gamma*a), alpha*c/(c+gamma*a)],
//   alpha = sgn(ab), beta = sgn(bc), gamma = sgn(ac), sgn(0) = +1;  accumulator cell (u,v) in [-1,1]^2 maps back
//   to the point [v, sgn(u)u + sgn(v)v - 1, u].
// Rasterisation as accumulate_lines (cht.cpp:163-197): steps = round(max|d|)+1, positions lo + j*(hi-lo)/(steps-1),
// rounded; votes are line lengths in 2^-16 fixed point (cht.cpp:194 ignores the weights; cht.h:18 wants them).
// Parity unpinned; validated on synthetic pencils with known vanishing points.
inline float sgn1(float x) { return x >= 0.f ? 1.f : -1.f; }

void cht_accumulate(const LinePencilModel& M, int d, std::vector<uint64_t>& acc) {
    acc.assign(size_t(d) * d, 0);
    const float sc = float(d - 1);
    for (int i = 0; i < M.size(); ++i) {
        const float a = M.h[i].x, b = M.h[i].y, c = M.h[i].z;
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
        const uint64_t vote = uint64_t(M.length[i] * 65536.0f + 0.5f);
        for (int s = 0; s < 3; ++s) {
            if (!ok[s] || !ok[s + 1]) continue;
            const float x0 = std::round((px[s] + 1.f) * 0.5f * sc), y0 = std::round((py[s] + 1.f) * 0.5f * sc);
            const float x1 = std::round((px[s + 1] + 1.f) * 0.5f * sc), y1 = std::round((py[s + 1] + 1.f) * 0.5f * sc);
            if (!(x0 >= 0.f && x0 <= sc && y0 >= 0.f && y0 <= sc && x1 >= 0.f && x1 <= sc && y1 >= 0.f && y1 <= sc)) continue;
            const int steps = int(std::round(std::max(std::fabs(x1 - x0), std::fabs(y1 - y0)))) + 1;
            const float sx = steps > 1 ? (x1 - x0) / float(steps - 1) : 0.f, sy = steps > 1 ? (y1 - y0) / float(steps - 1) : 0.f;
            for (int j = 0; j < steps; ++j) {
                const int xi = int(std::round(x0 + float(j) * sx)), yi = int(std::round(y0 + float(j) * sy));
                acc[size_t(yi) * d + xi] += vote;
            }
        }
    }
}

V3 cht_peak(const std::vector<uint64_t>& acc, int d) {
    size_t best = 0;
    for (size_t i = 1; i < acc.size(); ++i)
        if (acc[i] > acc[best]) best = i;  // first maximum in row-major order
    const int iy = int(best / d), ix = int(best % d);
    const float u = float(ix) / float(d - 1) * 2.f - 1.f, v = float(iy) / float(d - 1) * 2.f - 1.f;
    return {v, sgn1(u) * u + sgn1(v) * v - 1.f, u};
}

// The accumulator as an estimator (cht.h:13-24 + the loop of estimator.h:99-145; parity unpinned): every round
// accumulates the votes of the lines still in the game FROM SCRATCH (the product takes the removed lines' votes back
// out of one accumulator instead, cht.h:18 -- integer votes make the two identical), takes the strongest cell as the
// hypothesis, refits on the remaining lines within tol of it (estimator.h:74-76) and applies the usual verdicts.
void estimate_line_pencils_cht(std::vector<LineSegment>& lines, int max_models, float inlier_deg, float garbage_deg, int d,
                               std::vector<V3>* models_out, std::vector<uint32_t>* cells_out) {
    BBox bb = bounding_box(lines);
    V2 p0 = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    LinePencilModel model(normalize_lines(lines, p0, scale));
    const float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    const int N = model.size();
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int num_observations = N, k = 0;
    while (num_observations >= 2 && k < max_models) {
        std::vector<int> obs;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0) obs.push_back(i);
        std::vector<LineSegment> sub_lines;  // (the accumulator only reads h and length: a sub-model of the observations)
        LinePencilModel sub = model;
        sub.h.clear(); sub.anchor.clear(); sub.direction.clear(); sub.length.clear();
        for (int i : obs) {
            sub.h.push_back(model.h[i]);
            sub.anchor.push_back(model.anchor[i]);
            sub.direction.push_back(model.direction[i]);
            sub.length.push_back(model.length[i]);
        }
        std::vector<uint64_t> acc;
        cht_accumulate(sub, d, acc);
        size_t best = 0;
        for (size_t i = 1; i < acc.size(); ++i)
            if (acc[i] > acc[best]) best = i;
        const V3 p = cht_peak(acc, d);
        std::vector<int> inl;
        for (int i : obs)
            if (model.error1(p, i) < tol) inl.push_back(i);
        const V3 h = model.fit_optimal(inl);
        if (models_out) models_out->push_back(h);
        if (cells_out) cells_out->push_back(uint32_t(best));
        int n_in = 0, n_gb = 0;
        for (int i : obs) {
            const float e = model.error1(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
            }
        }
        num_observations -= n_in + n_gb;
        ++k;
    }
    for (int i = 0; i < N; ++i) lines[i].group_id = garbage_flag[i] == 1 ? -1 : inlier_flag[i];
}

// estimate_multiple_structures (estimator.h:99-145) driven by PROSAC instead of RANSAC
// estimator.h:82-96 — DirectEstimator (compiled header, never instantiated: parity unpinned): the lines whose Hough
// weight exceeds 0.95 (an empty set means all lines, line_pencil.cpp:114-117) decide the refit alone.
V3 direct_solve(const LinePencilModel& model, const std::vector<int>& indices) {
    std::vector<float> weights = get_weights_fixed(model, indices);
    std::vector<int> inl;
    for (size_t j = 0; j < indices.size(); ++j)
        if (weights[j] > 0.95f) inl.push_back(indices[j]);
    return model.fit_optimal(inl);
}

void estimate_line_pencils_direct(std::vector<LineSegment>& lines, int max_models, float inlier_deg, float garbage_deg) {
    BBox bb = bounding_box(lines);
    V2 p = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    LinePencilModel model(normalize_lines(lines, p, scale));
    float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    int N = model.size();
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int num_observations = N, k = 0;
    while (num_observations >= 2 && k < max_models) {  // estimator.h:99-145 around the direct solver
        std::vector<int> obs;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0) obs.push_back(i);
        V3 h = direct_solve(model, obs);
        int n_in = 0, n_gb = 0;
        for (int i : obs) {
            float e = model.error1(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
            }
        }
        num_observations -= n_in + n_gb;
        ++k;
    }
    for (int i = 0; i < N; ++i) lines[i].group_id = garbage_flag[i] == 1 ? -1 : inlier_flag[i];
}

void estimate_line_pencils_prosac(std::vector<LineSegment>& lines, int max_models, float inlier_deg, float garbage_deg,
                                  const ProsacParams& P, uint64_t seed, std::vector<ProsacTrace>* traces = nullptr) {
    BBox bb = bounding_box(lines);
    V2 p = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    LinePencilModel model(normalize_lines(lines, p, scale));
    float tol = cos_threshold(inlier_deg), garbage_tol = cos_threshold(garbage_deg);
    int N = model.size();
    std::vector<int> inlier_flag(N, -1), garbage_flag(N, 0);
    int num_observations = N, k = 0;
    while (num_observations >= 2 && k < max_models) {
        std::vector<int> obs;
        for (int i = 0; i < N; ++i)
            if (inlier_flag[i] < 0 && garbage_flag[i] == 0) obs.push_back(i);
        ProsacTrace tr;
        V3 h = prosac_solve(model, obs, tol, P, seed, uint32_t(k), &tr);
        if (traces) traces->push_back(tr);
        int n_in = 0, n_gb = 0;
        for (int i : obs) {
            float e = model.error1(h, i);
            if (e < tol) {
                inlier_flag[i] = k;
                ++n_in;
            } else if (e >= tol && e < garbage_tol) {
                garbage_flag[i] = 1;
                ++n_gb;
            }
        }
        num_observations -= n_in + n_gb;
        ++k;
    }
    for (int i = 0; i < N; ++i) lines[i].group_id = garbage_flag[i] == 1 ? -1 : inlier_flag[i];
}

// ---------------------------------------------------------------------------------------
// transform.cpp
V3 fit_single_vanishing_points(const std::vector<LineSegment>& lines, int g) {  // :24-47
    BBox bb = bounding_box(lines);
    V2 c = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    LinePencilModel model(normalize_lines(lines, c, scale));
    std::vector<int> idx;
    if (g > 0)  // note: g > 0, so group 0 also means "all lines" (transform.cpp:35)
        for (size_t i = 0; i < lines.size(); ++i)
            if (lines[i].group_id == g) idx.push_back(int(i));
    V3 vp = normalize_point(model.fit_optimal(idx));
    if (vp.z > 0) {
        vp.x = scale * vp.x + c.x;
        vp.y = scale * vp.y + c.y;
    }
    return vp;
}

std::map<int, V3> fit_vanishing_points(const std::vector<LineSegment>& lines) {  // :52-81
    BBox bb = bounding_box(lines);
    V2 c = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    LinePencilModel model(normalize_lines(lines, c, scale));
    std::set<int> groups;
    for (auto& l : lines) groups.insert(l.group_id);
    groups.erase(-1);
    std::map<int, V3> res;
    for (int g : groups) {
        std::vector<int> idx;
        for (size_t i = 0; i < lines.size(); ++i)
            if (lines[i].group_id == g) idx.push_back(int(i));
        V3 vp = normalize_point(model.fit_optimal(idx));
        if (vp.z > 0) {
            vp.x = scale * vp.x + c.x;
            vp.y = scale * vp.y + c.y;
        }
        res[g] = vp;
    }
    return res;
}

inline void mat3_mul(const float A[9], const float B[9], float C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) C[i * 3 + j] = (A[i * 3 + 0] * B[0 * 3 + j] + A[i * 3 + 1] * B[1 * 3 + j]) + A[i * 3 + 2] * B[2 * 3 + j];
}
inline V3 mat3_vec(const float A[9], V3 v) {
    return {(A[0] * v.x + A[1] * v.y) + A[2] * v.z, (A[3] * v.x + A[4] * v.y) + A[5] * v.z, (A[6] * v.x + A[7] * v.y) + A[8] * v.z};
}
inline void mat3_inverse(const float m[9], float inv[9]) {  // cofactor form, as Eigen's fixed-size 3x3 inverse
    auto M = [&](int i, int j) { return m[(i % 3) * 3 + (j % 3)]; };
    auto cof = [&](int i, int j) { return M(i + 1, j + 1) * M(i + 2, j + 2) - M(i + 1, j + 2) * M(i + 2, j + 1); };
    float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    float id = 1.0f / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) inv[i * 3 + j] = cof(j, i) * id;
}

// transform.cpp:84-133.  out: 4 rows (TL,TR,BR,BL) x 3.
void compute_image_transform(int width, int height, V3 hvp, V3 vvp, float out[12]) {
    V3 vl = cross(hvp, vvp);
    float H[9] = {1, 0, 0, 0, 1, 0, vl.x / vl.z, vl.y / vl.z, vl.z / vl.z};
    V3 vh = mat3_vec(H, hvp);
    V3 vv = mat3_vec(H, vvp);
    if (vh.x < 0) vh = {-vh.x, -vh.y, -vh.z};
    if (vv.y < 0) vv = {-vv.x, -vv.y, -vv.z};
    V2 a0 = normalized2({vh.x, vh.y});
    V2 a1 = normalized2({vv.x, vv.y});
    float A1[9] = {a0.x, a1.x, 0, a0.y, a1.y, 0, 0, 0, 1};
    float A[9], M[9];
    mat3_inverse(A1, A);
    mat3_mul(A, H, M);
    float cx[4] = {0.f, float(width), float(width), 0.f};
    float cy[4] = {0.f, 0.f, float(height), float(height)};
    for (int k = 0; k < 4; ++k) {
        V3 c{cx[k] - float(width) / 2, cy[k] - float(height) / 2, 1.f};
        V3 wv = mat3_vec(M, c);
        float s = 1.0f / wv.z;  // (1.0/row(2).array()): the double literal is narrowed to the array's float scalar
        out[k * 3 + 0] = wv.x * s + float(width) / 2;
        out[k * 3 + 1] = wv.y * s + float(height) / 2;
        out[k * 3 + 2] = wv.z * s;
    }
}

V3 select_vertical_point(const std::vector<V3>& vps, V3 center, float angular_tolerance, float min_distance) {  // :136-170
    float cos_thr = std::cos(angular_tolerance / 180.0f * float(M_PI));
    for (auto& v : vps) {
        V2 d = direction(v, center);
        bool angular = std::fabs(d.x * 0.f + d.y * 1.f) > cos_thr;
        bool dist = distance(v, center) > min_distance;
        if (angular && dist) return v;
    }
    return {0, 1, 0};
}

V3 select_horizontal_point(const std::vector<V3>& vps, V3 center, V3 vertical, float min_distance) {  // :173-211
    V2 vd = direction(vertical, center);
    for (auto& v : vps) {
        if (v.x == vertical.x && v.y == vertical.y && v.z == vertical.z) continue;
        V2 d = direction(v, center);
        float angle_score = d.x * vd.x + d.y * vd.y;
        bool horizon = angle_score < 0.05 && angle_score > -0.7;
        bool dist = distance(v, center) > min_distance;
        if (horizon && dist) return v;
    }
    return {1, 0, 0};
}

inline Point pt(V3 v) { return {v.x, v.y, v.z}; }

// interface.cpp:122-208
ImageTransform compute_rectification_transform(const LineSegment* lines, int n_lines, int width, int height,
                                               const RectificationConfig& cfg) {
    ImageTransform T;
    T.width = width;
    T.height = height;
    std::vector<V3> vps;
    if (n_lines > 0) {
        std::vector<LineSegment> groupped(lines, lines + n_lines);
        for (auto& kv : fit_vanishing_points(groupped)) vps.push_back(kv.second);
    }
    V3 image_center{float(width) / 2, float(height) / 2, 1};
    float diagonal_size = std::sqrt(image_center.x * image_center.x + image_center.y * image_center.y);
    float min_v = std::max(cfg.vertical_vp_min_distance, 1.0f) * diagonal_size;
    V3 vp_v = select_vertical_point(vps, image_center, cfg.vertical_vp_angular_tolerance, min_v);
    float min_h = std::max(cfg.horizontal_vp_min_distance, 1.0f) * diagonal_size;
    V3 vp_h = select_horizontal_point(vps, image_center, vp_v, min_h);
    V3 v1 = vp_h;
    if (v1.z != 0) {
        v1.x -= image_center.x;
        v1.y -= image_center.y;
    }
    V3 v2 = vp_v;
    if (v2.z != 0) {
        v2.x -= image_center.x;
        v2.y -= image_center.y;
    }
    V3 v1_hat = v1;
    switch (cfg.h_strategy) {
        case ROTATE_H: v1_hat.z = 0; break;
        case ROTATE_V: v1_hat = {-v2.y, v2.x, 0}; break;
        case RECTIFY: break;
        case KEEP:
        default: v1_hat = {1, 0, 0}; break;
    }
    V3 v2_hat = v2;
    switch (cfg.v_strategy) {
        case ROTATE_H: v2_hat = {-v1.y, v1.x, 0}; break;
        case ROTATE_V: v2_hat.z = 0; break;
        case RECTIFY: break;
        case KEEP:
        default: v2_hat = {0, 1, 0}; break;
    }
    float tf[12];
    compute_image_transform(width, height, v1_hat, v2_hat, tf);
    if (v1_hat.z != 0) {
        v1_hat.x += image_center.x;
        v1_hat.y += image_center.y;
    }
    if (v2_hat.z != 0) {
        v2_hat.x += image_center.x;
        v2_hat.y += image_center.y;
    }
    T.top_left = {tf[0], tf[1], tf[2]};
    T.top_right = {tf[3], tf[4], tf[5]};
    T.bottom_right = {tf[6], tf[7], tf[8]};
    T.bottom_left = {tf[9], tf[10], tf[11]};
    T.horizontal_vp = pt(v1_hat);
    T.vertical_vp = pt(v2_hat);
    return T;
}

// interface.cpp:93-119
ImageTransform compute_rectification_transform_from_vp(int width, int height, Point vp_h, Point vp_v) {
    V3 c{float(width) / 2, float(height) / 2, 0};
    V3 v1{vp_h.x, vp_h.y, vp_h.z};
    if (v1.z != 0) {
        v1.x -= c.x;
        v1.y -= c.y;
    }
    V3 v2{vp_v.x, vp_v.y, vp_v.z};
    if (v2.z != 0) {
        v2.x -= c.x;
        v2.y -= c.y;
    }
    float tf[12];
    compute_image_transform(width, height, v1, v2, tf);
    ImageTransform T;
    T.width = width;
    T.height = height;
    T.top_left = {tf[0], tf[1], tf[2]};
    T.top_right = {tf[3], tf[4], tf[5]};
    T.bottom_right = {tf[6], tf[7], tf[8]};
    T.bottom_left = {tf[9], tf[10], tf[11]};
    T.horizontal_vp = vp_h;
    T.vertical_vp = vp_v;
    return T;
}

// interface.cpp:218-265
void assign_to_group(const LineSegment* lines_array, int n_lines, LineSegment* new_lines, int n_new, float tol_deg) {
    std::vector<LineSegment> lines(lines_array, lines_array + n_lines);
    auto g2vp = fit_vanishing_points(lines);
    std::vector<float> best(n_new, 0.f);
    std::vector<int> best_idx(n_new, 0);
    for (auto& kv : g2vp) {
        for (int i = 0; i < n_new; ++i) {
            const auto& l = new_lines[i];
            float ax = (l.x2 + l.x1) / 2, ay = (l.y2 + l.y1) / 2;
            float dx = l.x2 - l.x1, dy = l.y2 - l.y1;
            float len = std::sqrt(dx * dx + dy * dy);
            float x = inclination1(ax, ay, dx / len, dy / len, kv.second);
            if (x > best[i]) {
                best[i] = x;
                best_idx[i] = kv.first;
            }
        }
    }
    float thr = std::cos(tol_deg / 180 * float(M_PI));
    for (int i = 0; i < n_new; ++i)
        if (best[i] > thr) new_lines[i].group_id = best_idx[i];
}

// interface.cpp:26-32
std::vector<LineSegment> filter_lines(const std::vector<LineSegment>& lines, float min_length) {
    min_length = std::max(min_length, LINE_MIN_LENGTH);
    std::vector<LineSegment> out;
    for (auto& l : lines)
        if (seg_length(l) > min_length && l.err < LINE_MAX_ERR) out.push_back(l);
    return out;
}

// image.cpp:11-19
std::vector<float> image_from_buffer(const float* buffer, int width, int height, int stride) {
    if (stride < 0) {
        buffer = buffer + std::ptrdiff_t(height - 1) * stride;
        stride = -stride;
    }
    std::vector<float> im(size_t(width) * height);
    for (int r = 0; r < height; ++r) std::memcpy(&im[size_t(r) * width], buffer + std::ptrdiff_t(r) * stride, sizeof(float) * width);
    return im;
}

// ---------------------------------------------------------------------------------------
// refine: line_detector.cpp:254-444
LineSegment merge_lines(const std::vector<LineSegment>& lines) {  // :254-274
    if (lines.size() == 1) return lines[0];
    size_t n = 2 * lines.size();
    std::vector<float> xr(n), xc(n), W(n);
    float wsum = 0.f, lsum = 0.f;
    for (size_t i = 0; i < lines.size(); ++i) {
        const auto& ln = lines[i];
        float l = seg_length(ln);
        float wt = l * ln.weight;
        xr[2 * i + 0] = ln.y1;
        xc[2 * i + 0] = ln.x1;
        xr[2 * i + 1] = ln.y2;
        xc[2 * i + 1] = ln.x2;
        W[2 * i + 0] = wt;
        W[2 * i + 1] = wt;
        wsum = wsum + wt;
        lsum = lsum + l;
    }
    LineSegment merged = fit_line_parameters(xr.data(), xc.data(), W.data(), n);
    merged.weight = wsum / lsum;
    return merged;
}

// The four constants of the pair test (line_detector.cpp:369,382,385): today's values are the defaults; the sweep of
// tools/sweep_refine_pins.py varies them to look for the older set the doc/ artefacts were made with.
struct RefineParams {
    // (doubles, as the reference's literals: a float value is compared after promotion, and 0.02f < 0.02)
    double cos_gate = 0.99;    // |d_i . d_j| below this: not parallel enough           (:369)
    double max_offset = 0.02;  // normal offset of both endpoints, in units of the longer segment's length (:382)
    double lo = -0.5, hi = 1.5;  // overlap window along the longer segment (:385)
    // Segments no longer than this take no part in the pair graph (they pass through unmerged).  Today's reference has no
    // such gate (0); the version that made the doc/ artefacts behaves as if it had one near 6 px (tools/sweep_refine_pins.py:
    // 791 -> 831 of the 848 golden rows).
    double min_pair_length = 0.0;
};
std::vector<LineSegment> postprocess_lines_segments(const std::vector<LineSegment>& lines, const ThreadContext& ctx,
                                                    const RefineParams& P = RefineParams()) {  // :332-444
    int n = int(lines.size());
    std::vector<V2> d(n), nn(n);
    std::vector<float> l(n);
    for (int i = 0; i < n; ++i) {
        float dx = lines[i].x2 - lines[i].x1, dy = lines[i].y2 - lines[i].y1;
        float len = std::sqrt(dx * dx + dy * dy);
        d[i] = {dx / len, dy / len};
        l[i] = len;
        nn[i] = {-d[i].y, d[i].x};
    }
    std::vector<std::vector<int>> adj(n);  // aff(i,j)=1 for i<j
#pragma omp parallel for schedule(dynamic, 1) num_threads(ctx.get()) if (ctx.enabled())
    for (int i = 0; i < n; ++i) {
        const auto& li = lines[i];
        if (!(double(l[i]) > P.min_pair_length)) continue;
        for (int j = i + 1; j < n; ++j) {
            const auto& lj = lines[j];
            if (!(double(l[j]) > P.min_pair_length)) continue;
            if (std::fabs(d[i].x * d[j].x + d[i].y * d[j].y) < P.cos_gate) continue;
            float w00, w01, w10, w11;  // W(row, col): rows = the two endpoints, col0 = along, col1 = normal
            if (l[i] < l[j]) {
                float ax = li.x1 - lj.x1, ay = li.y1 - lj.y1, bx = li.x2 - lj.x1, by = li.y2 - lj.y1;
                w00 = (ax * d[j].x + ay * d[j].y) / l[j];
                w01 = (ax * nn[j].x + ay * nn[j].y) / l[j];
                w10 = (bx * d[j].x + by * d[j].y) / l[j];
                w11 = (bx * nn[j].x + by * nn[j].y) / l[j];
            } else {
                float ax = lj.x1 - li.x1, ay = lj.y1 - li.y1, bx = lj.x2 - li.x1, by = lj.y2 - li.y1;
                w00 = (ax * d[i].x + ay * d[i].y) / l[i];
                w01 = (ax * nn[i].x + ay * nn[i].y) / l[i];
                w10 = (bx * d[i].x + by * d[i].y) / l[i];
                w11 = (bx * nn[i].x + by * nn[i].y) / l[i];
            }
            if (std::max(std::fabs(w01), std::fabs(w11)) < P.max_offset) {
                bool any_gt = (w00 > P.lo) || (w10 > P.lo);
                bool any_lt = (w00 < P.hi) || (w10 < P.hi);
                if (any_gt && any_lt) adj[i].push_back(j);
            }
        }
    }
    // graph_components + dfs (:277-329): forward-only BFS (u > n), label = first vertex
    std::vector<uint8_t> visited(n, 0);
    std::vector<int> comp(n, 0);
    for (int v = 0; v < n; ++v) {
        if (visited[v]) continue;
        std::queue<int> nodes;
        nodes.push(v);
        while (!nodes.empty()) {
            int u = nodes.front();
            nodes.pop();
            visited[u] = 1;
            comp[u] = v;
            for (int t : adj[u])
                if (!visited[t]) nodes.push(t);
        }
    }
    std::set<int> labels(comp.begin(), comp.end());
    std::vector<LineSegment> res;
    for (int lbl : labels) {
        std::vector<LineSegment> grp;
        for (int j = 0; j < n; ++j)
            if (comp[j] == lbl) grp.push_back(lines[j]);
        res.push_back(merge_lines(grp));
    }
    return res;
}

}  // namespace

// =======================================================================================
// C entry points (ctypes).  All arrays are caller-allocated.
extern "C" {

void orc_gauss_deriv_kernel(int size, float sigma, int dir_x, float* out) { gauss_deriv_kernel(size, sigma, dir_x != 0, out); }

void orc_bin_trig(int n_bins, float* st, float* ct) { bin_trig(n_bins, st, ct); }

// the reference's 25-tap correlation (second path, see conv_gradients_25tap)
void orc_conv_gradients_25tap(const float* img, int w, int h, float* dx, float* dy) { conv_gradients_25tap(img, w, h, dx, dy); }

// dx, dy, mag: w*h floats; bin: w*h int32; dmask: w*h uint8 (bit b set iff the 3x3 dilation
// of grad_bin==b covers the pixel; 0 on the 1-px border) — the lazily evaluated form of the
// 8 masked planes that the GPU path stores.
void orc_filter_stage(const float* img, int w, int h, int num_threads, float* dx, float* dy, float* mag, int32_t* bin,
                      uint8_t* dmask, float* planes /* optional 8*w*h */) {
    ThreadContext ctx(num_threads);
    image_gradients(img, w, h, dx, dy, mag, ctx);
    std::vector<std::vector<float>> grad;
    gradient_directions(dx, dy, w, h, 8, bin, grad, ctx);
    size_t n = size_t(w) * h;
    if (dmask) {
        std::fill(dmask, dmask + n, uint8_t(0));
        for (int i = 1; i < h - 1; ++i)
            for (int j = 1; j < w - 1; ++j) {
                uint8_t m = 0;
                for (int a = -1; a <= 1; ++a)
                    for (int b = -1; b <= 1; ++b) m |= uint8_t(1u << bin[size_t(i + a) * w + (j + b)]);
                dmask[size_t(i) * w + j] = m;
            }
    }
    if (planes)
        for (int b = 0; b < 8; ++b) std::memcpy(planes + size_t(b) * n, grad[b].data(), n * sizeof(float));
}

// Seeds in canonical order.  Returns the count (may exceed cap; only cap are written).
int orc_find_seeds(const float* mag, const int32_t* bin, int w, int h, int32_t* rows, int32_t* cols, float* vals,
                   int32_t* bins, int cap, float* min_seed_value_out) {
    ThreadContext ctx(-1);
    size_t n = size_t(w) * h;
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::max(mx, mag[i]);
    float msv = mx * (1 - std::max(std::min(SEED_RATIO, 1.f), 0.f));
    if (min_seed_value_out) *min_seed_value_out = msv;
    auto seed = find_peaks(mag, w, h, SEED_DIST, msv, ctx);
    for (size_t i = 0; i < seed.size() && int(i) < cap; ++i) {
        rows[i] = seed[i].i;
        cols[i] = seed[i].j;
        vals[i] = seed[i].v;
        bins[i] = bin[size_t(seed[i].i) * w + seed[i].j];
    }
    return int(seed.size());
}

// Whole detector.  label (optional, w*h int32) = claiming seed index or -1; comp_seed
// (optional, cap ints) = seed index of each returned line.  tolerance lets the pin-2 test
// run at 0.3.  times (optional, 5 doubles, ms): gradients, directions, seeds, components, fitting.
int orc_find_line_segments(const float* img, int w, int h, float tolerance, int num_threads, LineSegment* out, int cap,
                           int32_t* label, int32_t* comp_seed, int* n_seeds, double* times) {
    ThreadContext ctx(num_threads);
    std::vector<PeakPoint> seeds;
    std::vector<int> cs;
    StageTimes st;
    auto lines = find_line_segments(img, w, h, SEED_DIST, SEED_RATIO, tolerance, ctx, label, &seeds, &cs, &st);
    for (size_t i = 0; i < lines.size() && int(i) < cap; ++i) {
        out[i] = lines[i];
        if (comp_seed) comp_seed[i] = cs[i];
    }
    if (n_seeds) *n_seeds = int(seeds.size());
    if (times) {
        times[0] = st.gradients;
        times[1] = st.directions;
        times[2] = st.seeds;
        times[3] = st.components;
        times[4] = st.fitting;
    }
    return int(lines.size());
}

void orc_fit_line_parameters(const float* xr, const float* xc, const float* w, int n, LineSegment* out) {
    *out = fit_line_parameters(xr, xc, w, size_t(n));
}

int orc_filter_lines(const LineSegment* in, int n, float min_length, LineSegment* out) {
    auto f = filter_lines(std::vector<LineSegment>(in, in + n), min_length);
    std::copy(f.begin(), f.end(), out);
    return int(f.size());
}

// refine with other constants than today's (experiment / pin sweep only)
int orc_refine_lines_params(const LineSegment* in, int n, double cos_gate, double max_offset, double lo, double hi, double min_pair_length,
                            LineSegment* out) {
    RefineParams P;
    P.min_pair_length = min_pair_length;
    P.cos_gate = cos_gate;
    P.max_offset = max_offset;
    P.lo = lo;
    P.hi = hi;
    auto r = postprocess_lines_segments(std::vector<LineSegment>(in, in + n), ThreadContext(-1), P);
    std::copy(r.begin(), r.end(), out);
    return int(r.size());
}

int orc_refine_lines(const LineSegment* in, int n, int num_threads, LineSegment* out) {
    auto r = postprocess_lines_segments(std::vector<LineSegment>(in, in + n), ThreadContext(num_threads));
    std::copy(r.begin(), r.end(), out);
    return int(r.size());
}

// group ids written in place; models (optional) receives up to max_models*3 floats (normalised space).
int orc_estimate_line_pencils(LineSegment* lines, int n, int max_models, float inlier_deg, float garbage_deg, int n_iter,
                              uint64_t seed, int num_threads, float* models) {
    std::vector<LineSegment> v(lines, lines + n);
    std::vector<V3> m;
    estimate_line_pencils(v, max_models, inlier_deg, garbage_deg, n_iter, seed, ThreadContext(num_threads), &m);
    std::copy(v.begin(), v.end(), lines);
    if (models)
        for (size_t i = 0; i < m.size(); ++i) {
            models[3 * i + 0] = m[i].x;
            models[3 * i + 1] = m[i].y;
            models[3 * i + 2] = m[i].z;
        }
    return int(m.size());
}

// One RANSAC solve over `indices` of the pencil model of already-normalised lines: returns the
// raw best hypothesis, its score and iteration (for the scoring-kernel parity test).
void orc_ransac_best(const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float tol, int n_iter,
                     uint64_t seed, uint32_t round, int num_threads, float* best_h, float* best_score, int* best_iter,
                     float* refit_h) {
    LinePencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
    std::vector<int> idx(indices, indices + n_idx);
    V3 raw;
    V3 r = ransac_solve(model, idx, tol, n_iter, seed, round, ThreadContext(num_threads), &raw, best_score, best_iter);
    best_h[0] = raw.x;
    best_h[1] = raw.y;
    best_h[2] = raw.z;
    if (refit_h) {
        refit_h[0] = r.x;
        refit_h[1] = r.y;
        refit_h[2] = r.z;
    }
}

void orc_sample_pair(uint64_t seed, uint32_t round, uint32_t iter, uint32_t n, uint32_t* a, uint32_t* b) {
    sample_pair(seed, round, iter, n, *a, *b);
}

// math_utils.cpp:14-39 verbatim semantics (kept for the distribution test of the canonical sampler)
void orc_choice_knuth_mt(uint32_t mt_seed, int N, int n, int n_draws, int32_t* out) {
    std::mt19937 rng(mt_seed);
    for (int d = 0; d < n_draws; ++d) {
        int t = 0, m = 0;
        std::uniform_real_distribution<float> uniform(0, 1);
        while (m < n) {
            double u = uniform(rng);
            if ((N - t) * u >= n - m)
                t++;
            else {
                out[d * n + m] = t;
                t++;
                m++;
            }
        }
    }
}

float orc_cos_threshold(float deg) { return cos_threshold(deg); }

void orc_pencil_model(const LineSegment* lines_norm, int n, float* h3, float* anchor2, float* dir2, float* length) {
    LinePencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
    for (int i = 0; i < n; ++i) {
        h3[3 * i] = model.h[i].x;
        h3[3 * i + 1] = model.h[i].y;
        h3[3 * i + 2] = model.h[i].z;
        anchor2[2 * i] = model.anchor[i].x;
        anchor2[2 * i + 1] = model.anchor[i].y;
        dir2[2 * i] = model.direction[i].x;
        dir2[2 * i + 1] = model.direction[i].y;
        length[i] = model.length[i];
    }
}

void orc_normalize_lines(const LineSegment* lines, int n, LineSegment* out, float* center2, float* scale) {
    std::vector<LineSegment> v(lines, lines + n);
    BBox bb = bounding_box(v);
    V2 p = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float s = std::max(sz.x, sz.y);
    auto nl = normalize_lines(v, p, s);
    std::copy(nl.begin(), nl.end(), out);
    if (center2) {
        center2[0] = p.x;
        center2[1] = p.y;
    }
    if (scale) *scale = s;
}

void orc_get_weights(const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* out) {
    LinePencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
    auto w = model.get_weights(std::vector<int>(indices, indices + n_idx));
    std::copy(w.begin(), w.end(), out);
}

// returns number of groups; ids[k], vps[3k..]
int orc_fit_vanishing_points(const LineSegment* lines, int n, int32_t* ids, float* vps, int cap) {
    auto m = fit_vanishing_points(std::vector<LineSegment>(lines, lines + n));
    int k = 0;
    for (auto& kv : m) {
        if (k < cap) {
            ids[k] = kv.first;
            vps[3 * k] = kv.second.x;
            vps[3 * k + 1] = kv.second.y;
            vps[3 * k + 2] = kv.second.z;
        }
        ++k;
    }
    return k;
}

void orc_fit_vanishing_point(const LineSegment* lines, int n, int group, Point* out) {
    V3 v = fit_single_vanishing_points(std::vector<LineSegment>(lines, lines + n), group);
    *out = pt(v);
}

void orc_assign_to_group(const LineSegment* lines, int n, LineSegment* new_lines, int n_new, float tol_deg) {
    assign_to_group(lines, n, new_lines, n_new, tol_deg);
}

void orc_compute_rectification_transform(const LineSegment* lines, int n, int width, int height,
                                         const RectificationConfig* cfg, ImageTransform* out) {
    *out = compute_rectification_transform(lines, n, width, height, *cfg);
}

void orc_compute_rectification_transform_from_vp(int width, int height, const Point* vp_h, const Point* vp_v,
                                                 ImageTransform* out) {
    *out = compute_rectification_transform_from_vp(width, height, *vp_h, *vp_v);
}

// interface.cpp:35-80 with an explicit RANSAC seed.  Returns n (0 on the reference's nullptr paths).
int orc_find_line_segment_groups(const float* buffer, int width, int height, int stride, float min_length, int refine,
                                 int num_threads, uint64_t seed, LineSegment* out, int cap, double* times /*7: 5 stages, ransac, total*/) {
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    ThreadContext ctx(num_threads);
    auto im = image_from_buffer(buffer, width, height, stride);
    StageTimes st;
    auto lines = find_line_segments(im.data(), width, height, SEED_DIST, SEED_RATIO, TRACE_TOLERANCE, ctx, nullptr,
                                    nullptr, nullptr, &st);
    if (lines.size() < 2) return 0;
    if (refine) lines = postprocess_lines_segments(lines, ctx);
    auto filtered = filter_lines(lines, min_length);
    if (filtered.empty()) return 0;
    auto t1 = clk::now();
    estimate_line_pencils(filtered, MAX_MODELS, ESTIMATOR_INLIER_MAX_ANGLE_DEG, ESTIMATOR_GARBAGE_MAX_ANGLE_DEG,
                          RANSAC_MAX_ITER, seed, ctx);
    auto t2 = clk::now();
    for (size_t i = 0; i < filtered.size() && int(i) < cap; ++i) out[i] = filtered[i];
    if (times) {
        times[0] = st.gradients;
        times[1] = st.directions;
        times[2] = st.seeds;
        times[3] = st.components;
        times[4] = st.fitting;
        times[5] = std::chrono::duration<double, std::milli>(t2 - t1).count();
        times[6] = std::chrono::duration<double, std::milli>(t2 - t0).count();
    }
    return int(filtered.size());
}

void orc_get_weights_fixed(const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* out) {
    LinePencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
    auto w = get_weights_fixed(model, std::vector<int>(indices, indices + n_idx));
    std::copy(w.begin(), w.end(), out);
}

// PROSAC over `indices` of the pencil model of already-normalised lines; trace4 = iterations, n_star, best_iter, I_N_best
void orc_prosac_solve(const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float tol, int T_N,
                      uint64_t seed, uint32_t round, float* h3, int32_t* trace4) {
    LinePencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
    ProsacParams P;
    P.T_N = T_N;
    ProsacTrace tr;
    V3 h = prosac_solve(model, std::vector<int>(indices, indices + n_idx), tol, P, seed, round, &tr);
    h3[0] = h.x;
    h3[1] = h.y;
    h3[2] = h.z;
    if (trace4) {
        trace4[0] = tr.iterations;
        trace4[1] = tr.n_star;
        trace4[2] = tr.best_iter;
        trace4[3] = tr.I_N_best;
    }
}

void orc_direct_solve(const LineSegment* lines_norm, int n, const int32_t* indices, int n_idx, float* h3) {
    LinePencilModel model(std::vector<LineSegment>(lines_norm, lines_norm + n));
    V3 h = direct_solve(model, std::vector<int>(indices, indices + n_idx));
    h3[0] = h.x;
    h3[1] = h.y;
    h3[2] = h.z;
}

int orc_estimate_line_pencils_direct(LineSegment* lines, int n, int max_models, float inlier_deg, float garbage_deg) {
    std::vector<LineSegment> v(lines, lines + n);
    estimate_line_pencils_direct(v, max_models, inlier_deg, garbage_deg);
    std::copy(v.begin(), v.end(), lines);
    return 0;
}

int orc_estimate_line_pencils_prosac(LineSegment* lines, int n, int max_models, float inlier_deg, float garbage_deg,
                                     int T_N, uint64_t seed) {
    std::vector<LineSegment> v(lines, lines + n);
    ProsacParams P;
    P.T_N = T_N;
    estimate_line_pencils_prosac(v, max_models, inlier_deg, garbage_deg, P, seed);
    std::copy(v.begin(), v.end(), lines);
    return 0;
}

// lines in image coordinates; returns the de-normalised vanishing point (z = 0 for an ideal point) and, optionally, the accumulator
void orc_cht_vanishing_point(const LineSegment* lines, int n, int d, float* vp3, uint64_t* acc_out) {
    std::vector<LineSegment> v(lines, lines + n);
    BBox bb = bounding_box(v);
    V2 c = bbox_center(bb);
    V2 sz = bbox_size(bb);
    float scale = std::max(sz.x, sz.y);
    LinePencilModel model(normalize_lines(v, c, scale));
    std::vector<uint64_t> acc;
    cht_accumulate(model, d, acc);
    if (acc_out) std::copy(acc.begin(), acc.end(), acc_out);
    V3 p = normalize_point(cht_peak(acc, d));
    if (p.z > 0) {
        p.x = scale * p.x + c.x;
        p.y = scale * p.y + c.y;
    }
    vp3[0] = p.x;
    vp3[1] = p.y;
    vp3[2] = p.z;
}

// returns the number of rounds; models3: max_models x 3 floats, cells: max_models words (both optional)
int orc_estimate_line_pencils_cht(LineSegment* lines, int n, int max_models, float inlier_deg, float garbage_deg, int d,
                                  float* models3, uint32_t* cells) {
    std::vector<LineSegment> v(lines, lines + n);
    std::vector<V3> models;
    std::vector<uint32_t> cs;
    estimate_line_pencils_cht(v, max_models, inlier_deg, garbage_deg, d, &models, &cs);
    std::copy(v.begin(), v.end(), lines);
    for (size_t k = 0; k < models.size(); ++k) {
        if (models3) {
            models3[3 * k + 0] = models[k].x;
            models3[3 * k + 1] = models[k].y;
            models3[3 * k + 2] = models[k].z;
        }
        if (cells) cells[k] = cs[k];
    }
    return int(models.size());
}

int orc_niter_ransac(double p, double eps, int s, int nmax) { return niter_RANSAC(p, eps, s, nmax); }

int orc_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

}  // extern "C"
