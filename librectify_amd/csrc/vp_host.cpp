#include "vp_host.h"

#include <algorithm>
#include <cmath>
#include <queue>
#include <set>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace lramd {
namespace {

constexpr float kEpsH = 1e-6f;       // reference config.h:10
constexpr float kLineMaxErrH = 2.0f;  // config.h:50
constexpr float kLineMinLenH = 5.f;   // config.h:56

inline Vec3 cross3(const Vec3& a, const Vec3& b) {
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}

// Eigen's MatrixBase::normalized(): leaves a zero vector alone
inline Vec3 unit3(const Vec3& v) {
    const float z = (v.x * v.x + v.y * v.y) + v.z * v.z;
    if (z > 0.0f) {
        const float n = std::sqrt(z);
        return {v.x / n, v.y / n, v.z / n};
    }
    return v;
}
inline Vec2 unit2(const Vec2& v) {
    const float z = v.x * v.x + v.y * v.y;
    if (z > 0.0f) {
        const float n = std::sqrt(z);
        return {v.x / n, v.y / n};
    }
    return v;
}

// geometry.cpp:232-238
inline Vec3 normalize_point(const Vec3& p) {
    if (std::fabs(p.z) < kEpsH) return {p.x, p.y, 0.f};
    return {p.x / p.z, p.y / p.z, 1.f};
}
// geometry.cpp:240-245
inline Vec2 direction_to(const Vec3& a, const Vec3& b) { return unit2({a.x - b.x * a.z, a.y - b.y * a.z}); }
// geometry.cpp:247-255
inline float distance_to(const Vec3& a, const Vec3& b) {
    if (a.z < kEpsH || b.z < kEpsH) return INFINITY;
    const float dx = a.x - b.x, dy = a.y - b.y;
    return std::sqrt(dx * dx + dy * dy);
}

// geometry.cpp:214-229 for a single (anchor, unit direction) pair.  The row-wise normalisation
// there has no zero guard, so a zero vector yields NaN (and compares false everywhere).
inline float inclination(const Vec2& a, const Vec2& d, const Vec3& p) {
    float vx, vy;
    if (std::fabs(p.z) < kEpsH) {
        vx = p.x;
        vy = p.y;
    } else {
        const float qx = p.x / p.z, qy = p.y / p.z;
        vx = qx - a.x;
        vy = qy - a.y;
    }
    const float nn = vx * vx + vy * vy;
    const float nrm = std::sqrt(nn);
    const float ux = vx / nrm, uy = vy / nrm;
    return std::fabs(ux * d.x + uy * d.y);
}

// Smallest-eigenvalue eigenvector of a symmetric 3x3 (float in, cyclic Jacobi in double).
// Stands in for Eigen::SelfAdjointEigenSolver<Matrix3f> (line_pencil.cpp:123-127).
Vec3 smallest_eigenvector(const float c[9]) {
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
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double cs = 1.0 / std::sqrt(t * t + 1.0), sn = t * cs;
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
    const double nn = std::sqrt(nx * nx + ny * ny + nz * nz);
    return {(float)(nx / nn), (float)(ny / nn), (float)(nz / nn)};
}

// 2x2 major axis for the refine merge (same closed form as the GPU fit kernel)
void major_axis(float a_, float b_, float c_, float& d_r, float& d_c) {
    const double a = a_, b = b_, c = c_;
    const double hd = (a - c) * 0.5;
    const double rad = std::sqrt(hd * hd + b * b);
    const double lmax = (a + c) * 0.5 + rad;
    double vr, vc;
    if (a >= c) {
        vr = lmax - c;
        vc = b;
    } else {
        vr = b;
        vc = lmax - a;
    }
    const double nn = std::sqrt(vr * vr + vc * vc);
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

// canonical reduction tree (64 strided partials + xor butterfly), host form
template <class F>
float tree_sum(size_t n, F term) {
    float lane[64];
    for (int j = 0; j < 64; ++j) {
        float acc = 0.0f;
        for (size_t i = (size_t)j; i < n; i += 64) acc = acc + term(i);
        lane[j] = acc;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        float nxt[64];
        for (int j = 0; j < 64; ++j) nxt[j] = lane[j] + lane[j ^ off];
        for (int j = 0; j < 64; ++j) lane[j] = nxt[j];
    }
    return lane[0];
}

// geometry.cpp:20-61 on host (used by the refine merge only; the detector's fit is on the GPU)
LineSegment fit_line_host(const std::vector<float>& xr, const std::vector<float>& xc, const std::vector<float>& w) {
    const size_t n = w.size();
    const float S = tree_sum(n, [&](size_t i) { return w[i]; });
    std::vector<float> wn(n);
    for (size_t i = 0; i < n; ++i) wn[i] = w[i] / S;
    const float a_r = tree_sum(n, [&](size_t i) { return wn[i] * xr[i]; });
    const float a_c = tree_sum(n, [&](size_t i) { return wn[i] * xc[i]; });
    std::vector<float> cr(n), cc(n);
    for (size_t i = 0; i < n; ++i) {
        cr[i] = xr[i] - a_r;
        cc[i] = xc[i] - a_c;
    }
    const float crr = tree_sum(n, [&](size_t i) { return (cr[i] * wn[i]) * cr[i]; });
    const float crc = tree_sum(n, [&](size_t i) { return (cr[i] * wn[i]) * cc[i]; });
    const float ccc = tree_sum(n, [&](size_t i) { return (cc[i] * wn[i]) * cc[i]; });
    float d_r, d_c;
    major_axis(crr, crc, ccc, d_r, d_c);
    const float n_r = -d_c, n_c = d_r;
    float t0 = INFINITY, t1 = -INFINITY;
    for (size_t i = 0; i < n; ++i) {
        const float t = cr[i] * d_r + cc[i] * d_c;
        t0 = std::min(t0, t);
        t1 = std::max(t1, t);
    }
    const float es = tree_sum(n, [&](size_t i) { return std::fabs(cr[i] * n_r + cc[i] * n_c); });
    LineSegment l;
    l.x1 = a_c + d_c * t0;
    l.y1 = a_r + d_r * t0;
    l.x2 = a_c + d_c * t1;
    l.y2 = a_r + d_r * t1;
    l.weight = S / (float)n;
    l.err = es / (float)n;
    l.group_id = -1;
    return l;
}

inline void mul33(const float A[9], const float B[9], float C[9]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = (A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j]) + A[i * 3 + 2] * B[6 + j];
}
inline Vec3 mul3v(const float A[9], const Vec3& v) {
    return {(A[0] * v.x + A[1] * v.y) + A[2] * v.z, (A[3] * v.x + A[4] * v.y) + A[5] * v.z,
            (A[6] * v.x + A[7] * v.y) + A[8] * v.z};
}
// cofactor inverse, the form Eigen uses for fixed 3x3
inline void inv33(const float m[9], float inv[9]) {
    auto M = [&](int i, int j) { return m[(i % 3) * 3 + (j % 3)]; };
    auto cof = [&](int i, int j) { return M(i + 1, j + 1) * M(i + 2, j + 2) - M(i + 1, j + 2) * M(i + 2, j + 1); };
    const float c0 = cof(0, 0), c1 = cof(1, 0), c2 = cof(2, 0);
    const float det = (c0 * m[0] + c1 * m[3]) + c2 * m[6];
    const float id = 1.0f / det;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) inv[i * 3 + j] = cof(j, i) * id;
}

// transform.cpp:84-133; rows TL, TR, BR, BL
void image_transform(int width, int height, const Vec3& hvp, const Vec3& vvp, float out[12]) {
    const Vec3 vl = cross3(hvp, vvp);
    const float H[9] = {1, 0, 0, 0, 1, 0, vl.x / vl.z, vl.y / vl.z, vl.z / vl.z};
    Vec3 vh = mul3v(H, hvp), vv = mul3v(H, vvp);
    if (vh.x < 0) vh = {-vh.x, -vh.y, -vh.z};
    if (vv.y < 0) vv = {-vv.x, -vv.y, -vv.z};
    const Vec2 a0 = unit2({vh.x, vh.y}), a1 = unit2({vv.x, vv.y});
    const float A1[9] = {a0.x, a1.x, 0, a0.y, a1.y, 0, 0, 0, 1};
    float A[9], M[9];
    inv33(A1, A);
    mul33(A, H, M);
    const float cx[4] = {0.f, (float)width, (float)width, 0.f};
    const float cy[4] = {0.f, 0.f, (float)height, (float)height};
    for (int k = 0; k < 4; ++k) {
        const Vec3 c{cx[k] - (float)width / 2, cy[k] - (float)height / 2, 1.f};
        const Vec3 wv = mul3v(M, c);
        const float s = 1.0f / wv.z;
        out[k * 3 + 0] = wv.x * s + (float)width / 2;
        out[k * 3 + 1] = wv.y * s + (float)height / 2;
        out[k * 3 + 2] = wv.z * s;
    }
}

// transform.cpp:136-170
Vec3 pick_vertical(const std::vector<Vec3>& vps, const Vec3& center, float tol_deg, float min_distance) {
    const float cos_thr = std::cos(tol_deg / 180.0f * (float)M_PI);
    for (const Vec3& v : vps) {
        const Vec2 d = direction_to(v, center);
        const bool angular = std::fabs(d.x * 0.f + d.y * 1.f) > cos_thr;
        if (angular && distance_to(v, center) > min_distance) return v;
    }
    return {0, 1, 0};
}
// transform.cpp:173-211
Vec3 pick_horizontal(const std::vector<Vec3>& vps, const Vec3& center, const Vec3& vertical, float min_distance) {
    const Vec2 vd = direction_to(vertical, center);
    for (const Vec3& v : vps) {
        if (v.x == vertical.x && v.y == vertical.y && v.z == vertical.z) continue;
        const Vec2 d = direction_to(v, center);
        const float score = d.x * vd.x + d.y * vd.y;
        const bool horizon = score < 0.05 && score > -0.7;
        if (horizon && distance_to(v, center) > min_distance) return v;
    }
    return {1, 0, 0};
}

inline Point to_point(const Vec3& v) { return Point{v.x, v.y, v.z}; }

void fill_corners(ImageTransform& T, const float tf[12]) {
    T.top_left = {tf[0], tf[1], tf[2]};
    T.top_right = {tf[3], tf[4], tf[5]};
    T.bottom_right = {tf[6], tf[7], tf[8]};
    T.bottom_left = {tf[9], tf[10], tf[11]};
}

}  // namespace

PencilModel::PencilModel(const std::vector<LineSegment>& lines) {
    const size_t n = lines.size();
    h.resize(n);
    anchor.resize(n);
    direction.resize(n);
    length.resize(n);
    for (size_t i = 0; i < n; ++i) {
        const LineSegment& l = lines[i];
        h[i] = unit3(cross3({l.x1, l.y1, 1.f}, {l.x2, l.y2, 1.f}));  // geometry.cpp:64-69
        anchor[i] = {(l.x2 + l.x1) / 2, (l.y2 + l.y1) / 2};          // geometry.cpp:72-75
        const float dx = l.x2 - l.x1, dy = l.y2 - l.y1;              // geometry.cpp:78-81
        const float len = std::sqrt(dx * dx + dy * dy);
        length[i] = len;
        direction[i] = {dx / len, dy / len};
    }
}

Vec3 PencilModel::fit(int a, int b) const { return cross3(h[a], h[b]); }

bool PencilModel::sample_check(int a, int b) const {
    const float dx = h[a].x - h[b].x, dy = h[a].y - h[b].y, dz = h[a].z - h[b].z;
    return std::sqrt((dx * dx + dy * dy) + dz * dz) > degeneracy_tol;
}

Vec3 PencilModel::fit_optimal(const std::vector<int>& idx) const {
    // cov = sum_i (h_i * len_i) h_i^T over the index set in its order (empty set: every line), each of the nine sums
    // with the canonical tree T() -- what the device's peeling kernel computes (kernels_groups.hip)
    const size_t m = idx.empty() ? (size_t)size() : idx.size();
    float cov[9];
    for (int a = 0; a < 3; ++a)
        for (int b = 0; b < 3; ++b)
            cov[a * 3 + b] = tree_sum(m, [&](size_t j) {
                const int i = idx.empty() ? (int)j : idx[j];
                const float hv[3] = {h[i].x, h[i].y, h[i].z};
                const float t = hv[a] * length[i];
                return t * hv[b];
            });
    return smallest_eigenvector(cov);
}

float PencilModel::error(const Vec3& hyp, int i) const { return -inclination(anchor[i], direction[i], hyp) + 1.0f; }

Normalisation bbox_normalisation(const std::vector<LineSegment>& lines) {
    float minx = INFINITY, miny = INFINITY, maxx = -INFINITY, maxy = -INFINITY;
    for (const LineSegment& l : lines) {
        minx = std::min(minx, std::min(l.x1, l.x2));
        miny = std::min(miny, std::min(l.y1, l.y2));
        maxx = std::max(maxx, std::max(l.x1, l.x2));
        maxy = std::max(maxy, std::max(l.y1, l.y2));
    }
    const float sx = maxx - minx, sy = maxy - miny;
    Normalisation n;
    n.center = {minx + 0.5f * sx, miny + 0.5f * sy};
    n.scale = std::max(sx, sy);
    return n;
}

std::vector<LineSegment> normalise(const std::vector<LineSegment>& lines, const Normalisation& nrm) {
    std::vector<LineSegment> out(lines);
    for (LineSegment& l : out) {
        l.x1 = (l.x1 - nrm.center.x) / nrm.scale;
        l.y1 = (l.y1 - nrm.center.y) / nrm.scale;
        l.x2 = (l.x2 - nrm.center.x) / nrm.scale;
        l.y2 = (l.y2 - nrm.center.y) / nrm.scale;
    }
    return out;
}

float cos_threshold(float deg) { return 1.0f - std::cos(deg / 180.f * (float)M_PI); }

float segment_length(const LineSegment& l) {
    const float dx = l.x2 - l.x1, dy = l.y2 - l.y1;
    return std::sqrt(dx * dx + dy * dy);
}

std::vector<LineSegment> filter_lines(const std::vector<LineSegment>& lines, float min_length) {
    min_length = std::max(min_length, kLineMinLenH);
    std::vector<LineSegment> out;
    out.reserve(lines.size());
    for (const LineSegment& l : lines)
        if (segment_length(l) > min_length && l.err < kLineMaxErrH) out.push_back(l);
    return out;
}

// graph_components + dfs + merge_lines (line_detector.cpp:254-329,409-443): forward-only BFS over the edges
// i -> j (j > i), label = first vertex, one merged segment per label in ascending label order.
static std::vector<LineSegment> merge_graph(const std::vector<LineSegment>& lines, const std::vector<std::vector<int>>& adj) {
    const int n = (int)lines.size();
    std::vector<uint8_t> visited(n, 0);
    std::vector<int> comp(n, 0);
    for (int v = 0; v < n; ++v) {
        if (visited[v]) continue;
        std::queue<int> nodes;
        nodes.push(v);
        while (!nodes.empty()) {
            const int u = nodes.front();
            nodes.pop();
            visited[u] = 1;
            comp[u] = v;
            for (int t : adj[u])
                if (!visited[t]) nodes.push(t);
        }
    }
    // members per label in ascending index order, labels in ascending order (what the reference's
    // std::set walk + linear scans produce, without their O(n * labels) cost)
    std::vector<std::vector<int>> members(n);
    for (int j = 0; j < n; ++j) members[comp[j]].push_back(j);
    std::vector<LineSegment> res;
    for (int lbl = 0; lbl < n; ++lbl) {
        if (members[lbl].empty()) continue;
        std::vector<const LineSegment*> grp;
        for (int j : members[lbl]) grp.push_back(&lines[j]);
        if (grp.size() == 1) {
            res.push_back(*grp[0]);
            continue;
        }
        // merge_lines (:254-274): endpoints weighted by length*weight
        const size_t m = 2 * grp.size();
        std::vector<float> xr(m), xc(m), W(m);
        float wsum = 0.f, lsum = 0.f;
        for (size_t i = 0; i < grp.size(); ++i) {
            const LineSegment& ln = *grp[i];
            const float l = segment_length(ln);
            const float wt = l * ln.weight;
            xr[2 * i] = ln.y1;
            xc[2 * i] = ln.x1;
            xr[2 * i + 1] = ln.y2;
            xc[2 * i + 1] = ln.x2;
            W[2 * i] = wt;
            W[2 * i + 1] = wt;
            wsum = wsum + wt;
            lsum = lsum + l;
        }
        LineSegment merged = fit_line_host(xr, xc, W);
        merged.weight = wsum / lsum;
        res.push_back(merged);
    }
    return res;
}


// line_detector.cpp:254-444 with the O(n^2) pair test on the host (used for small n; context.hip switches to
// refine_pairs_kernel for large n); the merge graph walk reproduces the reference's forward-only BFS (:293).
std::vector<LineSegment> refine_lines(const std::vector<LineSegment>& lines) {
    const int n = (int)lines.size();
    std::vector<Vec2> d(n), nv(n);
    std::vector<float> len(n);
    for (int i = 0; i < n; ++i) {
        const float dx = lines[i].x2 - lines[i].x1, dy = lines[i].y2 - lines[i].y1;
        const float l = std::sqrt(dx * dx + dy * dy);
        d[i] = {dx / l, dy / l};
        len[i] = l;
        nv[i] = {-d[i].y, d[i].x};
    }
    std::vector<std::vector<int>> adj(n);
    for (int i = 0; i < n; ++i) {
        const LineSegment& li = lines[i];
        for (int j = i + 1; j < n; ++j) {
            const LineSegment& lj = lines[j];
            if (std::fabs(d[i].x * d[j].x + d[i].y * d[j].y) < 0.99) continue;
            const bool i_short = len[i] < len[j];
            const LineSegment& ref = i_short ? lj : li;   // frame of the longer one
            const LineSegment& oth = i_short ? li : lj;
            const Vec2& rd = i_short ? d[j] : d[i];
            const Vec2& rn = i_short ? nv[j] : nv[i];
            const float rl = i_short ? len[j] : len[i];
            const float ax = oth.x1 - ref.x1, ay = oth.y1 - ref.y1, bx = oth.x2 - ref.x1, by = oth.y2 - ref.y1;
            const float w00 = (ax * rd.x + ay * rd.y) / rl, w01 = (ax * rn.x + ay * rn.y) / rl;
            const float w10 = (bx * rd.x + by * rd.y) / rl, w11 = (bx * rn.x + by * rn.y) / rl;
            if (std::max(std::fabs(w01), std::fabs(w11)) < 0.02) {
                const bool any_gt = (w00 > -0.5) || (w10 > -0.5);
                const bool any_lt = (w00 < 1.5) || (w10 < 1.5);
                if (any_gt && any_lt) adj[i].push_back(j);
            }
        }
    }
    return merge_graph(lines, adj);
}

void refine_segment_table(const std::vector<LineSegment>& lines, std::vector<float>& t) {
    const size_t n = lines.size();
    t.resize(7 * n);
    for (size_t i = 0; i < n; ++i) {
        const float dx = lines[i].x2 - lines[i].x1, dy = lines[i].y2 - lines[i].y1;
        const float l = std::sqrt(dx * dx + dy * dy);
        t[7 * i + 0] = lines[i].x1;
        t[7 * i + 1] = lines[i].y1;
        t[7 * i + 2] = lines[i].x2;
        t[7 * i + 3] = lines[i].y2;
        t[7 * i + 4] = dx / l;
        t[7 * i + 5] = dy / l;
        t[7 * i + 6] = l;
    }
}

std::vector<LineSegment> refine_lines_from_edges(const std::vector<LineSegment>& lines,
                                                 const std::vector<std::pair<uint32_t, uint32_t>>& edges) {
    std::vector<std::vector<int>> adj(lines.size());
    for (const auto& e : edges) adj[e.first].push_back((int)e.second);
    return merge_graph(lines, adj);
}

std::map<int, Vec3> fit_vanishing_points(const std::vector<LineSegment>& lines) {
    const Normalisation nrm = bbox_normalisation(lines);
    const PencilModel model(normalise(lines, nrm));
    std::set<int> groups;
    for (const LineSegment& l : lines) groups.insert(l.group_id);
    groups.erase(-1);
    std::map<int, Vec3> res;
    for (int g : groups) {
        std::vector<int> idx;
        for (size_t i = 0; i < lines.size(); ++i)
            if (lines[i].group_id == g) idx.push_back((int)i);
        Vec3 vp = normalize_point(model.fit_optimal(idx));
        if (vp.z > 0) {
            vp.x = nrm.scale * vp.x + nrm.center.x;
            vp.y = nrm.scale * vp.y + nrm.center.y;
        }
        res[g] = vp;
    }
    return res;
}

Vec3 fit_single_vanishing_point(const std::vector<LineSegment>& lines, int g) {
    const Normalisation nrm = bbox_normalisation(lines);
    const PencilModel model(normalise(lines, nrm));
    std::vector<int> idx;
    if (g > 0)  // the reference tests g > 0 (transform.cpp:35): group 0, like any negative id, means all lines
        for (size_t i = 0; i < lines.size(); ++i)
            if (lines[i].group_id == g) idx.push_back((int)i);
    Vec3 vp = normalize_point(model.fit_optimal(idx));
    if (vp.z > 0) {
        vp.x = nrm.scale * vp.x + nrm.center.x;
        vp.y = nrm.scale * vp.y + nrm.center.y;
    }
    return vp;
}

ImageTransform rectification_transform(const LineSegment* lines, int n, int width, int height,
                                       const RectificationConfig& cfg) {
    ImageTransform T;
    T.width = width;
    T.height = height;
    std::vector<Vec3> vps;
    if (n > 0) {  // n == 0 is undefined in the reference (bounding box of nothing); here: ideal points, identity corners
        const std::vector<LineSegment> v(lines, lines + n);
        for (const auto& kv : fit_vanishing_points(v)) vps.push_back(kv.second);
    }
    const Vec3 center{(float)width / 2, (float)height / 2, 1};
    const float diag = std::sqrt(center.x * center.x + center.y * center.y);
    const float min_v = std::max(cfg.vertical_vp_min_distance, 1.0f) * diag;
    const Vec3 vp_v = pick_vertical(vps, center, cfg.vertical_vp_angular_tolerance, min_v);
    const float min_h = std::max(cfg.horizontal_vp_min_distance, 1.0f) * diag;
    const Vec3 vp_h = pick_horizontal(vps, center, vp_v, min_h);
    Vec3 v1 = vp_h;
    if (v1.z != 0) {
        v1.x -= center.x;
        v1.y -= center.y;
    }
    Vec3 v2 = vp_v;
    if (v2.z != 0) {
        v2.x -= center.x;
        v2.y -= center.y;
    }
    Vec3 v1_hat = v1;
    switch ((int)cfg.h_strategy) {
        case librectify::ROTATE_H: v1_hat.z = 0; break;
        case librectify::ROTATE_V: v1_hat = {-v2.y, v2.x, 0}; break;
        case librectify::RECTIFY: break;
        default: v1_hat = {1, 0, 0}; break;
    }
    Vec3 v2_hat = v2;
    switch ((int)cfg.v_strategy) {
        case librectify::ROTATE_H: v2_hat = {-v1.y, v1.x, 0}; break;
        case librectify::ROTATE_V: v2_hat.z = 0; break;
        case librectify::RECTIFY: break;
        default: v2_hat = {0, 1, 0}; break;
    }
    float tf[12];
    image_transform(width, height, v1_hat, v2_hat, tf);
    if (v1_hat.z != 0) {
        v1_hat.x += center.x;
        v1_hat.y += center.y;
    }
    if (v2_hat.z != 0) {
        v2_hat.x += center.x;
        v2_hat.y += center.y;
    }
    fill_corners(T, tf);
    T.horizontal_vp = to_point(v1_hat);
    T.vertical_vp = to_point(v2_hat);
    return T;
}

ImageTransform rectification_transform_from_vp(int width, int height, const Point& vp_h, const Point& vp_v) {
    const Vec2 c{(float)width / 2, (float)height / 2};
    Vec3 v1{vp_h.x, vp_h.y, vp_h.z};
    if (v1.z != 0) {
        v1.x -= c.x;
        v1.y -= c.y;
    }
    Vec3 v2{vp_v.x, vp_v.y, vp_v.z};
    if (v2.z != 0) {
        v2.x -= c.x;
        v2.y -= c.y;
    }
    float tf[12];
    image_transform(width, height, v1, v2, tf);
    ImageTransform T;
    T.width = width;
    T.height = height;
    fill_corners(T, tf);
    T.horizontal_vp = vp_h;
    T.vertical_vp = vp_v;
    return T;
}

void assign_groups(const LineSegment* lines, int n, LineSegment* new_lines, int n_new, float tol_deg) {
    const std::vector<LineSegment> v(lines, lines + n);
    const auto g2vp = fit_vanishing_points(v);
    std::vector<float> best(n_new, 0.f);
    std::vector<int> best_id(n_new, 0);
    for (const auto& kv : g2vp) {
        for (int i = 0; i < n_new; ++i) {
            const LineSegment& l = new_lines[i];
            const Vec2 a{(l.x2 + l.x1) / 2, (l.y2 + l.y1) / 2};
            const float dx = l.x2 - l.x1, dy = l.y2 - l.y1;
            const float len = std::sqrt(dx * dx + dy * dy);
            const float x = inclination(a, {dx / len, dy / len}, kv.second);
            if (x > best[i]) {
                best[i] = x;
                best_id[i] = kv.first;
            }
        }
    }
    const float thr = std::cos(tol_deg / 180 * (float)M_PI);
    for (int i = 0; i < n_new; ++i)
        if (best[i] > thr) new_lines[i].group_id = best_id[i];
}

}  // namespace lramd
