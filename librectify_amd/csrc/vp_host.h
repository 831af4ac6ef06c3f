// Host-side geometry of the vanishing-point path: line-pencil model, optimal refit, transform.
// These are O(n_lines) or O(1) computations (microseconds); the per-hypothesis scoring they
// drive runs on the GPU (kernels_ransac.hip).  Arithmetic follows the canonical form in
// fp32 as in the reference, no FP contraction, eigen-solvers in double.
#pragma once
#include <cstdint>
#include <map>
#include <vector>

#include "../../include/librectify.h"

namespace lramd {

using librectify::ImageTransform;
using librectify::LineSegment;
using librectify::Point;
using librectify::RectificationConfig;

struct Vec2 {
    float x, y;
};
struct Vec3 {
    float x, y, z;
};

// reference line_pencil.h:12-34 / line_pencil.cpp:25-32
struct PencilModel {
    std::vector<Vec3> h;          // unit homogeneous line
    std::vector<Vec2> anchor;     // segment midpoint
    std::vector<Vec2> direction;  // unit direction p1->p2
    std::vector<float> length;
    float degeneracy_tol = 0.05f;

    explicit PencilModel(const std::vector<LineSegment>& lines_norm);
    int size() const { return (int)h.size(); }
    Vec3 fit(int a, int b) const;                          // line_pencil.cpp:101-108
    bool sample_check(int a, int b) const;                 // :89-98
    Vec3 fit_optimal(const std::vector<int>& idx) const;   // :111-128 (empty => all lines)
    float error(const Vec3& hyp, int i) const;             // :131-134
};

struct Normalisation {
    Vec2 center;
    float scale;
};
Normalisation bbox_normalisation(const std::vector<LineSegment>& lines);  // geometry.cpp:96-112,272-282
std::vector<LineSegment> normalise(const std::vector<LineSegment>& lines, const Normalisation& nrm);  // :258-270

float cos_threshold(float deg);  // line_pencil.cpp:143-146
float segment_length(const LineSegment& l);

std::vector<LineSegment> filter_lines(const std::vector<LineSegment>& lines, float min_length);  // interface.cpp:26-32
std::vector<LineSegment> refine_lines(const std::vector<LineSegment>& lines);  // line_detector.cpp:332-444
// same, with the merge-graph edges (i < j) already computed (by the GPU pair kernel)
std::vector<LineSegment> refine_lines_from_edges(const std::vector<LineSegment>& lines,
                                                 const std::vector<std::pair<uint32_t, uint32_t>>& edges);
// per-segment record of the pair test: {x1, y1, x2, y2, unit dx, unit dy, length}
void refine_segment_table(const std::vector<LineSegment>& lines, std::vector<float>& table7);

std::map<int, Vec3> fit_vanishing_points(const std::vector<LineSegment>& lines);  // transform.cpp:52-81
Vec3 fit_single_vanishing_point(const std::vector<LineSegment>& lines, int g);    // transform.cpp:24-47

ImageTransform rectification_transform(const LineSegment* lines, int n, int width, int height,
                                       const RectificationConfig& cfg);                        // interface.cpp:122-208
ImageTransform rectification_transform_from_vp(int width, int height, const Point& vp_h, const Point& vp_v);  // :93-119
void assign_groups(const LineSegment* lines, int n, LineSegment* new_lines, int n_new, float tol_deg);        // :218-265

}  // namespace lramd
