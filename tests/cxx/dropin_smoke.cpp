// A caller written against the reference's public header compiles and links against librectify_amd.so unchanged.
// Inputs are the ones of the reference's smoke program (src/test.cpp:19-39); answers derived in SURVEY.md §4.
#include <cstdio>

#include "librectify.h"

using namespace librectify;

int main() {
    LineSegment A = {0, 0, 10, 0, 1, 0, 10};
    LineSegment B = {10, 0, 8, 5, 1, 0, 1};
    LineSegment C = {8, 5, 2, 5, 1, 0, 10};
    LineSegment D = {2, 5, 0, 0, 1, 0, 1};
    LineSegment ls[4] = {A, B, C, D};
    Point vp1 = fit_vanishing_point(ls, 4, 1);
    Point vp2 = fit_vanishing_point(ls, 4, 10);
    LineSegment line = {5, 1, 5, 4, 1, 0, -1};
    assign_to_group(ls, 4, &line, 1, 10);
    RectificationConfig cfg;
    ImageTransform T = compute_rectification_transform(ls, 0, 640, 480, cfg);
    LineSegment* none = nullptr;
    release_line_segments(&none);
    std::printf("vp1 %.4f %.4f %.1f\n", vp1.x, vp1.y, vp1.z);
    std::printf("vp2z %.1f\n", vp2.z);
    std::printf("group %d\n", line.group_id);
    std::printf("identity %.3f %.3f %.3f %.3f\n", T.top_left.x, T.top_left.y, T.bottom_right.x, T.bottom_right.y);
    std::printf("sizes %zu %zu %zu %zu\n", sizeof(LineSegment), sizeof(Point), sizeof(ImageTransform), sizeof(RectificationConfig));
    return 0;
}
