// The caller's recipe around the C API, without OpenCV (behaviour of the reference demo,
// autorectify.cpp:56-68,113-148,322-376): gray conversion, scale by 1/256, area-averaging prescale to at most
// `--max-size` pixels on the long side, find_line_segment_groups with min_length = max(w,h)/100, endpoints scaled
// back, compute_rectification_transform with horizontal_vp_min_distance = 2, and the two CSV files the demo writes
// (<prefix>_lines.csv: x1,y1,x2,y2,weight,err,group_id per row; <prefix>_tform.csv: TL, TR, BL, BR, hvp, vvp).
//
//   rectify_recipe in.pgm|in.ppm out_prefix [--max-size N|fraction] [--refine] [--threads N]
//                  [--h-strategy rotate_h|rotate_v|rectify|keep] [--v-strategy ...]
//
// Input is a binary PGM (P5, 8 bit) or PPM (P6, 8 bit; converted with the usual integer luma weights
// (4899 R + 9617 G + 1868 B + 8192) >> 14).  Image decoding and warping stay with the caller's imaging library.
// Links against librectify_amd.so exactly like a program written for the reference.
#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <string>
#include <vector>

#include "librectify.h"

using namespace librectify;

namespace {

struct Gray {
    int w = 0, h = 0;
    std::vector<float> px;  // row-major
};

bool read_token(std::istream& f, std::string& tok) {
    tok.clear();
    int c;
    while ((c = f.get()) != EOF) {
        if (c == '#') {
            while ((c = f.get()) != EOF && c != '\n') {
            }
        } else if (!std::isspace(c)) {
            break;
        }
    }
    if (c == EOF) return false;
    do {
        tok.push_back((char)c);
        c = f.get();
    } while (c != EOF && !std::isspace(c));
    return true;
}

// 8-bit gray levels as floats in [0, 1): value / 256 (autorectify.cpp:119)
bool load_pnm(const std::string& path, Gray& g) {
    std::ifstream f(path, std::ios::binary);
    std::string magic, tw, th, tm;
    if (!f || !read_token(f, magic) || !read_token(f, tw) || !read_token(f, th) || !read_token(f, tm)) return false;
    const int ch = magic == "P5" ? 1 : (magic == "P6" ? 3 : 0);
    if (!ch || std::atoi(tm.c_str()) != 255) return false;
    g.w = std::atoi(tw.c_str());
    g.h = std::atoi(th.c_str());
    if (g.w <= 0 || g.h <= 0) return false;
    std::vector<uint8_t> raw((size_t)g.w * g.h * ch);
    f.read(reinterpret_cast<char*>(raw.data()), (std::streamsize)raw.size());
    if ((size_t)f.gcount() != raw.size()) return false;
    g.px.resize((size_t)g.w * g.h);
    for (size_t i = 0; i < g.px.size(); ++i) {
        int v = raw[i * ch];
        if (ch == 3) v = (4899 * raw[i * 3] + 9617 * raw[i * 3 + 1] + 1868 * raw[i * 3 + 2] + 8192) >> 14;
        g.px[i] = (float)v * (1.0f / 256.0f);
    }
    return true;
}

// One axis of an area-averaging downscale: destination sample i is the mean of the source interval
// [i*s, (i+1)*s), s = n_src / n_dst, partially covered source samples weighted by their overlap.
struct Span {
    int first;
    std::vector<float> wgt;
};
std::vector<Span> area_spans(int n_src, int n_dst) {
    std::vector<Span> spans(n_dst);
    const double s = (double)n_src / n_dst;
    for (int i = 0; i < n_dst; ++i) {
        const double lo = i * s, hi = std::min((double)n_src, (i + 1) * s);
        const int a = (int)std::floor(lo), b = std::min(n_src - 1, (int)std::ceil(hi) - 1);
        spans[i].first = a;
        for (int j = a; j <= b; ++j) {
            const double ov = std::min(hi, (double)j + 1) - std::max(lo, (double)j);
            spans[i].wgt.push_back((float)(std::max(0.0, ov) / s));
        }
    }
    return spans;
}

// The demo's prescale (autorectify.cpp:56-68): scale = min(max_size / max(w, h), 1); area interpolation.
Gray prescale(const Gray& in, int max_size, float& scale) {
    scale = std::min((float)max_size / (float)std::max(in.w, in.h), 1.0f);
    if (scale == 1.0f) return in;
    Gray out;
    out.w = std::max(1, (int)std::lround(in.w * (double)scale));
    out.h = std::max(1, (int)std::lround(in.h * (double)scale));
    const std::vector<Span> sx = area_spans(in.w, out.w), sy = area_spans(in.h, out.h);
    std::vector<float> tmp((size_t)in.h * out.w);
    for (int y = 0; y < in.h; ++y)
        for (int x = 0; x < out.w; ++x) {
            float acc = 0.f;
            const float* row = &in.px[(size_t)y * in.w + sx[x].first];
            for (size_t j = 0; j < sx[x].wgt.size(); ++j) acc += sx[x].wgt[j] * row[j];
            tmp[(size_t)y * out.w + x] = acc;
        }
    out.px.assign((size_t)out.w * out.h, 0.f);
    for (int y = 0; y < out.h; ++y)
        for (size_t j = 0; j < sy[y].wgt.size(); ++j) {
            const float wj = sy[y].wgt[j];
            const float* row = &tmp[(size_t)(sy[y].first + (int)j) * out.w];
            float* dst = &out.px[(size_t)y * out.w];
            for (int x = 0; x < out.w; ++x) dst[x] += wj * row[x];
        }
    return out;
}

bool parse_strategy(const std::string& s, RectificationStrategy& out) {
    if (s == "rotate_h") out = ROTATE_H;
    else if (s == "rotate_v") out = ROTATE_V;
    else if (s == "rectify") out = RECTIFY;
    else if (s == "keep") out = KEEP;
    else return false;
    return true;
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) {
        std::fprintf(stderr,
                     "usage: %s in.pgm|in.ppm out_prefix [--max-size N|fraction] [--refine] [--threads N]\n"
                     "          [--h-strategy rotate_h|rotate_v|rectify|keep] [--v-strategy ...]\n",
                     argv[0]);
        return 2;
    }
    float max_size = 1200.f;  // the demo's default
    bool refine = false;
    int threads = -1;
    RectificationConfig cfg;
    cfg.horizontal_vp_min_distance = 2;  // autorectify.cpp:347
    for (int i = 3; i < argc; ++i) {
        const std::string a = argv[i];
        const bool has_val = i + 1 < argc;
        if (a == "--refine") refine = true;
        else if (a == "--max-size" && has_val) max_size = (float)std::atof(argv[++i]);
        else if (a == "--threads" && has_val) threads = std::atoi(argv[++i]);
        else if (a == "--h-strategy" && has_val && parse_strategy(argv[i + 1], cfg.h_strategy)) ++i;
        else if (a == "--v-strategy" && has_val && parse_strategy(argv[i + 1], cfg.v_strategy)) ++i;
        else {
            std::fprintf(stderr, "unknown or incomplete option: %s\n", a.c_str());
            return 2;
        }
    }
    Gray full;
    if (!load_pnm(argv[1], full)) {
        std::fprintf(stderr, "cannot read %s (binary PGM/PPM, 8 bit, expected)\n", argv[1]);
        return 1;
    }
    // a value below 1 is a fraction of the long side (autorectify.cpp:121-125)
    const int max_px = max_size < 1.f ? (int)(std::max(full.w, full.h) * max_size) : (int)max_size;
    float scale = 1.f;
    Gray img = prescale(full, std::max(1, max_px), scale);

    int n = 0;
    LineSegment* lines = find_line_segment_groups(img.px.data(), img.w, img.h, img.w,
                                                  (float)std::max(img.w, img.h) / 100.0f, refine, threads, &n);
    for (int i = 0; i < n; ++i) {  // back to the coordinates of the full image
        lines[i].x1 /= scale;
        lines[i].y1 /= scale;
        lines[i].x2 /= scale;
        lines[i].y2 /= scale;
    }
    const ImageTransform t = compute_rectification_transform(lines, n, full.w, full.h, cfg);

    const std::string prefix = argv[2];
    std::ofstream lf(prefix + "_lines.csv");
    for (int i = 0; i < n; ++i) {
        const LineSegment& l = lines[i];
        lf << l.x1 << "," << l.y1 << "," << l.x2 << "," << l.y2 << "," << l.weight << "," << l.err << "," << l.group_id
           << "\n";
    }
    std::ofstream tf(prefix + "_tform.csv");
    tf << t.top_left.x << "," << t.top_left.y << "\n";
    tf << t.top_right.x << "," << t.top_right.y << "\n";
    tf << t.bottom_left.x << "," << t.bottom_left.y << "\n";
    tf << t.bottom_right.x << "," << t.bottom_right.y << "\n";
    tf << t.horizontal_vp.x << "," << t.horizontal_vp.y << "," << t.horizontal_vp.z << "\n";
    tf << t.vertical_vp.x << "," << t.vertical_vp.y << "," << t.vertical_vp.z << "\n";
    std::printf("%dx%d -> %dx%d (scale %g), %d segments, wrote %s_lines.csv and %s_tform.csv\n", full.w, full.h, img.w,
                img.h, (double)scale, n, prefix.c_str(), prefix.c_str());
    release_line_segments(&lines);
    return lines == nullptr ? 0 : 1;
}
