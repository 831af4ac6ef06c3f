/*
 * librectify C API — drop-in boundary of the MI355X build.
 *
 * Declares the same six extern "C" entry points, four POD structs and one enum as the
 * reference's public header (reference src/librectify.h:44-222), with identical names,
 * field order, sizes and calling convention, so a program compiled against the reference
 * header links against librectify_amd.so unchanged.  Behavioural contract per function is
 * cited below; INTEGRATION.md shows the one-line link change.
 *
 * Data model (reference src/librectify.h:9-18): GRAYSCALE image in a float array, `stride`
 * in ELEMENTS between rows, possibly negative (buffer=X,stride=S is equivalent to
 * buffer=X+S*(H-1),stride=-S).  x = column, y = row.
 */
#pragma once

/* (the library is built with -fvisibility=hidden: what is declared between here and the matching pop is what it exports) */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#ifdef __cplusplus
namespace librectify {
extern "C" {
#else
#include <stdbool.h>
#endif

/* reference src/librectify.h:44-54 — 28 bytes */
struct LineSegment {
    float x1, y1, x2, y2; /* endpoints */
    float weight;         /* mean edge response of the supporting pixels */
    float err;            /* mean |normal offset| of the supporting pixels */
    int group_id;         /* vanishing-point group, -1 = unassigned */
};

/* reference src/librectify.h:60-63 — homogeneous point, 12 bytes */
struct Point {
    float x, y, z;
};

/* reference src/librectify.h:79-86 — 80 bytes */
struct ImageTransform {
    int width;
    int height;
    struct Point top_left, top_right, bottom_left, bottom_right;
    struct Point horizontal_vp;
    struct Point vertical_vp;
};

/* reference src/librectify.h:126-132 */
enum RectificationStrategy {
    ROTATE_H,
    ROTATE_V,
    RECTIFY,
    KEEP,
};

/* reference src/librectify.h:137-150 — 20 bytes; defaults 40, 1.5, RECTIFY, 1.5, RECTIFY */
struct RectificationConfig {
#ifdef __cplusplus
    float vertical_vp_angular_tolerance{40};
    float vertical_vp_min_distance{1.5f};
    RectificationStrategy v_strategy{RECTIFY};
    float horizontal_vp_min_distance{1.5f};
    RectificationStrategy h_strategy{RECTIFY};
#else
    float vertical_vp_angular_tolerance;
    float vertical_vp_min_distance;
    enum RectificationStrategy v_strategy;
    float horizontal_vp_min_distance;
    enum RectificationStrategy h_strategy;
#endif
};

#ifdef __cplusplus
#define LR_CFG_REF const RectificationConfig&
#define LR_PT_REF const Point&
typedef float InputPixelType;
#else
#define LR_CFG_REF const struct RectificationConfig*
#define LR_PT_REF const struct Point*
typedef float InputPixelType;
typedef struct LineSegment LineSegment;
typedef struct ImageTransform ImageTransform;
typedef struct Point Point;
typedef struct RectificationConfig RectificationConfig;
#endif

/*
 * Detect line segments and group them by vanishing point.
 * Replaces reference src/librectify.h:111-116 / src/interface.cpp:35-80.
 * Returns a new[]-allocated array (free with release_line_segments) or NULL with
 * *n_lines = 0 when fewer than 2 raw segments were found or none survived the
 * length/err filter.  num_threads < 0 = serial host stages (reference threading.h:24-27);
 * the per-pixel stages always run on the GPU.
 */
LineSegment* find_line_segment_groups(InputPixelType* buffer, int width, int height, int stride, float min_length,
                                      bool refine, int num_threads, int* n_lines);

/* Replaces reference src/librectify.h:123 / src/interface.cpp:268-275 (delete[] + null; null-safe). */
void release_line_segments(LineSegment** lines);

/* Replaces reference src/librectify.h:175-178 / src/interface.cpp:122-208. */
ImageTransform compute_rectification_transform(LineSegment* lines, int n_lines, int width, int height, LR_CFG_REF cfg);

/* Replaces reference src/librectify.h:181-183 / src/interface.cpp:93-119. */
ImageTransform compute_rectification_transform_from_vp(int width, int height, LR_PT_REF vp_h, LR_PT_REF vp_v);

/* Replaces reference src/librectify.h:199-201 / src/interface.cpp:211-215 (group <= 0 fits ALL lines: transform.cpp:35). */
Point fit_vanishing_point(const LineSegment* lines, int n_lines, int group);

/* Replaces reference src/librectify.h:219-222 / src/interface.cpp:218-265 (writes new_lines_array[i].group_id). */
void assign_to_group(const LineSegment* lines_array, int n_lines, LineSegment* new_lines_array, int n_new_lines,
                     float angular_tolarance);

#ifdef __cplusplus
} /* extern "C" */
} /* namespace librectify */
#endif

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
