// Stage 1 — fused edge filter for gfx950.
//
// One pass over the image produces everything the later stages need, so that the eight
// directional planes, the magnitude plane, the max-filtered plane and the eight dilated
// masks of the reference (line_detector.cpp:41-49,126-182; filter.cpp:29-98,161-168) are
// never materialised in HBM:
//
//   read  img                       4 B/px
//   write dx, dy                    8 B/px   (fp32, reused by flood + line fit)
//   write dmask                     1 B/px   (bit b: pixel lies in dilate3x3(grad_bin == b))
//   write peak candidates           sparse   (mag == max5x5 && mag > 0, with value, index, bin)
//   write per-band max(mag)         4 B/band
//
// Row streaming, one pixel per lane: a wavefront owns a band of 56 columns (lanes 4..59; four halo lanes on each
// side: conv radius 2 + NMS radius 2) x 30 rows and walks down it one image row per step.  Horizontal neighbours
// come from DPP wavefront shifts, vertical ones from registers that roll with the rows; there is no LDS and no
// barrier.  Arithmetic is the canonical form shared with the CPU oracle: the separable evaluation
// of the 5x5 Gaussian-derivative correlation (row pass hx, hs; column pass dx, dy; explicit fmaf),
// mag = sqrtf(dx*dx + dy*dy) without contraction, bin = first strict argmax of |fmaf(dx, sin, dy*cos)|.
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace lramd {
namespace {

// 1-D factors of the taps (filter.cpp:72-75): Hx(i,j) = d(x_j) g(y_i), Hy(i,j) = d(y_i) g(x_j) with
// d(t) = t/a exp(-t^2/2s^2), g(t) = exp(-t^2/2s^2); d(-t) = -d(t), d(0) = 0, g(-t) = g(t), g(0) = 1 exactly.
struct FilterTaps {
    float d1, d2, g1, g2;
    float st[kBins];
    float ct[kBins];
};

#ifndef LR_BAND_ROWS
#define LR_BAND_ROWS 30
#endif
constexpr int kBandRows = LR_BAND_ROWS;  // 2160 = 72 x 30: no ragged last band at 4K
constexpr int kBandSteps = kBandRows + 8;  // image rows y0-4 .. y0+33

// value of the lower / upper neighbour lane as a DPP wavefront shift (a VALU move, not an LDS crossbar
// trip like ds_bpermute); the end lanes read 0, and they are halo lanes anyway
__device__ __forceinline__ float from_lower(float v) {
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0x138 /* wave_shr:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ float from_upper(float v) {
    const int i = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_mov_dpp(i, 0x130 /* wave_shl:1 */, 0xF, 0xF, true));
}
__device__ __forceinline__ uint32_t from_lower(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x138, 0xF, 0xF, true);
}
__device__ __forceinline__ uint32_t from_upper(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, 0x130, 0xF, 0xF, true);
}

constexpr int kLaneCols = 56;

struct Roll1 {
    float hx[5];       // row pass of image rows t-4..t (slot = row mod 5): derivative along x ...
    float hs[5];       // ... and smoothing along x
    float mag[5];
    uint32_t bits[5];  // 1 << bin
    float q[10];       // prefetched image rows (nine steps ahead: loads share the in-order vmcnt with the stores)
};

// Unconditional load from a clamped position: pixels outside the image are never used by a valid output
// (conv_2d's zero border covers every window that would touch them), so any in-bounds value will do, and
// a branch-free load keeps the memory counter exact.  Row pointer is wave-uniform (SGPR base + lane offset).
__device__ __forceinline__ float load_px(const float* __restrict__ img, int stride, int h, int yr, int xcl) {
    const int yy = min(max(yr, 0), h - 1);
    const float* __restrict__ row = img + (size_t)yy * stride;
    // scalar row base + zero-extended 32-bit byte offset of the lane: one global_load with an SGPR base
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + (unsigned)xcl * 4u);
}

// Buffer-resource addressing (T8): 32-bit lane byte offset in a VGPR + scalar row byte offset, no per-access
// 64-bit address arithmetic on the vector ALU.  The descriptors are built once from wave-uniform values.
using BufRsrc = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ BufRsrc make_rsrc(const void* p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float load_px_in(const float* __restrict__ img, int stride, int yr, int xcl) {
    const float* __restrict__ row = img + (size_t)yr * stride;
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(row) + (unsigned)xcl * 4u);
}

template <int K10>
__device__ __forceinline__ void lane_step(Roll1& R, const int t, const float* __restrict__ img, const int w, const int h,
                                          const int stride, const FilterTaps& fc, float* __restrict__ dx_out,
                                          float* __restrict__ dy_out, uint8_t* __restrict__ dmask_out,
                                          uint64_t* __restrict__ cand_band, uint32_t& ncand, float& lmax, const int y0,
                                          const int x, const int xcl, const int lane, const bool useful,
                                          const BufRsrc r_img, const BufRsrc r_dx, const BufRsrc r_dy,
                                          const BufRsrc r_dm) {
    constexpr int K = K10 % 5;
    if (t >= kBandSteps) return;  // wave-uniform (the unrolled loop runs in chunks of ten steps)
    const float cur = R.q[K10];
    {
        const int yy = min(max(y0 - 4 + t + 9, 0), h - 1);  // clamped: see load_px
        R.q[(K10 + 9) % 10] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                  r_img, (unsigned)xcl * 4u, (unsigned)yy * (unsigned)stride * 4u, 0));
    }
    // row pass of the new image row (oracle: conv_gradients)
    const float l1 = from_lower(cur), r1 = from_upper(cur);
    const float l2 = from_lower(l1), r2 = from_upper(r1);
    R.hx[K] = fmaf(r2 - l2, fc.d2, (r1 - l1) * fc.d1);
    R.hs[K] = fmaf(r2 + l2, fc.g2, fmaf(r1 + l1, fc.g1, cur));
    if (t < 4) return;  // wave-uniform: five rows are not there yet

    const int yc = y0 - 6 + t;  // conv row: the middle one of the five
    // column pass: rows yc-2 .. yc+2 sit in slots K+1 .. K+5 (mod 5)
    const float ax = fmaf(R.hx[K] + R.hx[(K + 1) % 5], fc.g2,
                          fmaf(R.hx[(K + 4) % 5] + R.hx[(K + 2) % 5], fc.g1, R.hx[(K + 3) % 5]));
    const float ay = fmaf(R.hs[K] - R.hs[(K + 1) % 5], fc.d2, (R.hs[(K + 4) % 5] - R.hs[(K + 2) % 5]) * fc.d1);
    const bool ok = (yc >= 2) && (yc < h - 2) && (x >= 2) && (x < w - 2);  // conv_2d's zero border (filter.cpp:89-97)
    const float vx = ok ? ax : 0.f;
    const float vy = ok ? ay : 0.f;
    // (Round 5, measured and not kept -- profiles/r05_filter_variants.txt: the SQUARED magnitude in the rolling rows, with a
    // root only where a value is stored and where two squares are within a rounding interval of each other (exact: the
    // correctly rounded root is monotone): 40.4 us against 40.6 -- the root was never on the critical path -- and 45.1 us with
    // the tie test behind a wave-uniform branch of its own: a branch a step is worth 4-5 us of this kernel.  The arg-max
    // below as a three-level tournament: its seven comparison masks live in SGPR pairs across 38 unrolled steps, 106
    // scalar registers, 44 of them spilled, seven waves a SIMD instead of eight: 48 us.)
    R.mag[K] = sqrtf(vx * vx + vy * vy);
    // "first strict argmax from 0" == lowest bin attaining the maximum (all-zero responses give bin 0 both ways)
    float g[kBins];
#pragma unroll
    for (int b = 0; b < kBins; ++b) g[b] = directional(vx, vy, fc.st[b], fc.ct[b]);

    const float gm = fmaxf(fmaxf(fmaxf(g[0], g[1]), fmaxf(g[2], g[3])), fmaxf(fmaxf(g[4], g[5]), fmaxf(g[6], g[7])));
    uint32_t bit = 1u << 7;
#pragma unroll
    for (int b = kBins - 2; b >= 0; --b) bit = (g[b] == gm) ? (1u << b) : bit;
    R.bits[K] = bit;
    const bool in_img = useful && x < w;
    if (t >= 6 && t < 6 + kBandRows && yc < h) {  // yc in [y0, y0+32): wave-uniform, compile-time in t
        if (in_img) {
            const unsigned row = (unsigned)yc * (unsigned)w * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vx), r_dx, (unsigned)x * 4u, row, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vy), r_dy, (unsigned)x * 4u, row, 0);
        }
    }
    if (t < 8) return;  // wave-uniform

    const int yo = yc - 2;  // output row of the NMS / dilation
    if (yo >= h) return;  // wave-uniform
    const float cm = fmaxf(fmaxf(fmaxf(R.mag[0], R.mag[1]), fmaxf(R.mag[2], R.mag[3])), R.mag[4]);
    const float cl1 = from_lower(cm), cr1 = from_upper(cm);
    const float cl2 = from_lower(cl1), cr2 = from_upper(cr1);
    const float mx = fmaxf(fmaxf(fmaxf(cl2, cl1), fmaxf(cm, cr1)), cr2);
    const float c = R.mag[(K + 3) % 5];
    const uint32_t cb = R.bits[(K + 3) % 5];
    const uint32_t v = R.bits[(K + 2) % 5] | cb | R.bits[(K + 4) % 5];
    uint32_t dm = from_lower(v) | v | from_upper(v);
    if (yo == 0 || yo == h - 1 || x == 0 || x == w - 1) dm = 0;  // binary_dilate's 1-px zero border (filter.cpp:52-61)
    if (in_img) {
        lmax = fmaxf(lmax, c);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)dm, r_dm, (unsigned)x, (unsigned)yo * (unsigned)w, 0);
    }
    const bool peak = in_img && (yo >= 2) && (yo < h - 2) && (x >= 2) && (x < w - 2) && (c > 0.f) && (c == mx);
    const uint64_t m = __ballot(peak);
    if (m) {  // wave-uniform, rare
        if (peak) {
            const uint32_t idx = (uint32_t)yo * (uint32_t)w + (uint32_t)x;
            const uint32_t pb = (uint32_t)__ffs((int)cb) - 1u;
            cand_band[ncand + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] =
                ((uint64_t)__float_as_uint(c) << 32) | (uint64_t)((idx << 3) | pb);
        }
        ncand += (uint32_t)__popcll(m);
    }
}

__global__ __launch_bounds__(256) void filter_lanes_kernel(const float* __restrict__ img, int w, int h, int stride,
                                                           FilterTaps fc, float* __restrict__ dx_out,
                                                           float* __restrict__ dy_out, uint8_t* __restrict__ dmask_out,
                                                           uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_count,
                                                           uint32_t* __restrict__ tile_max, int bands_x, int band_begin,
                                                           int band_end) {
    const int lane = threadIdx.x & 63;
    // the band index is the same in all 64 lanes: say so, and every row pointer, row test and loop bound below
    // lives in SGPRs (scalar base + 32-bit lane offset addressing) instead of 64-bit VGPR arithmetic
#ifndef LR_NO_XCD_ORDER
    // Workgroups go to the eight XCDs round-robin, and each XCD has its own L2.  Give every XCD one contiguous run of
    // bands (a horizontal stripe of the image) instead of every eighth group of four: neighbouring bands share the halo
    // rows and columns they read, and the 128-byte lines of the byte mask they write, and now find them in their L2.
    const int chunk = (int)gridDim.x / 8;  // the launcher rounds the grid up to a multiple of eight
    const int wg = (int)(blockIdx.x & 7u) * chunk + (int)(blockIdx.x >> 3);
#else
    const int wg = (int)blockIdx.x;
#endif
    // (a launch covers the bands [band_begin, band_end): all of a frame, or the rows of it that have arrived -- see
    // launch_filter_rows)
    const int band = __builtin_amdgcn_readfirstlane(band_begin + wg * 4 + (int)(threadIdx.x >> 6));
    if (band >= band_end) return;
    const int by = band / bands_x, bx = band - by * bands_x;
    const int y0 = by * kBandRows;
    const int x = bx * kLaneCols - 4 + lane;
    const int xcl = min(max(x, 0), w - 1);
    const bool useful = lane >= 4 && lane <= 59;
    Roll1 R;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        R.hx[i] = 0.f;
        R.hs[i] = 0.f;
        R.mag[i] = 0.f;
        R.bits[i] = 1u;
    }
#pragma unroll
    for (int i = 0; i < 10; ++i) R.q[i] = (i < 9) ? load_px(img, stride, h, y0 - 4 + i, xcl) : 0.f;
    uint64_t* cand_band = cand + (size_t)band * (kLaneCols * kBandRows);
    uint32_t ncand = 0;
    float lmax = 0.f;
    const uint32_t npx = (uint32_t)w * (uint32_t)h;
    const BufRsrc r_img = make_rsrc(img, ((uint32_t)(h - 1) * (uint32_t)stride + (uint32_t)w) * 4u);
    const BufRsrc r_dx = make_rsrc(dx_out, npx * 4u), r_dy = make_rsrc(dy_out, npx * 4u), r_dm = make_rsrc(dmask_out, npx);
#define LR_STEP(k) lane_step<k>(R, t0 + k, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, x, xcl, lane, useful, r_img, r_dx, r_dy, r_dm)
    // fully unrolled: across a loop back-edge the compiler can only wait for vmcnt(0), which would expose the
    // latency of every store in flight once per iteration.
    // (An instantiation without border tests, clamps and predicates for the bands away from the image border -- some 55
    // scalar instructions per row fewer -- was measured twice, with the 25-tap and with the separable filter: no gain
    // either time, twice the code.  Removed in round 3.)
#pragma unroll
    for (int t0 = 0; t0 < (kBandSteps + 9) / 10 * 10; t0 += 10) {
        LR_STEP(0); LR_STEP(1); LR_STEP(2); LR_STEP(3); LR_STEP(4);
        LR_STEP(5); LR_STEP(6); LR_STEP(7); LR_STEP(8); LR_STEP(9);
    }
#undef LR_STEP
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) {
        tile_max[band] = __float_as_uint(lmax);
        cand_count[band] = ncand;
    }
}

}  // namespace

FilterGeom filter_geometry(int w, int h) {
    FilterGeom g;
    g.n_tiles = ((w + kLaneCols - 1) / kLaneCols) * ((h + kBandRows - 1) / kBandRows);
    g.cand_cap = kLaneCols * kBandRows;
    return g;
}

static int filter_taps(const FilterConsts& fc, FilterTaps& ft) {
    // the (anti)symmetry the separable form relies on is a property of the host's libm results: verify, never assume
    if (!(fc.d[2] == 0.f && fc.g[2] == 1.f && fc.d[1] == -fc.d[3] && fc.d[0] == -fc.d[4] && fc.g[1] == fc.g[3] &&
          fc.g[0] == fc.g[4])) {
        set_error("launch_filter: derivative factors are not (anti)symmetric on this host");
        return 1;
    }
    ft.d1 = fc.d[3];
    ft.d2 = fc.d[4];
    ft.g1 = fc.g[3];
    ft.g2 = fc.g[4];
    for (int b = 0; b < kBins; ++b) {
        ft.st[b] = fc.st[b];
        ft.ct[b] = fc.ct[b];
    }
    return 0;
}

int filter_band_rows() { return kBandRows; }
// a band walks kBandSteps image rows from kBandRows * by - 4 on (lane_step: row y0 - 4 + t, t < kBandSteps)
int filter_band_last_row(int by) { return kBandRows * by - 4 + kBandSteps - 1; }

// The bands whose image rows lie in [row_begin, row_end) -- band row `by` reads the image rows 30 by - 4 .. 30 by + 33 --
// i.e. the band rows [by_begin, by_end).  A frame that is still arriving over the link is filtered in a few such
// launches, each as soon as its rows are on the device (context.hip: find_groups_host).
int launch_filter_rows(const float* img, int w, int h, int stride, const FilterConsts& fc, float* dx, float* dy,
                       uint8_t* dmask, uint64_t* cand, uint32_t* cand_count, uint32_t* tile_max, int by_begin, int by_end,
                       hipStream_t s) {
    if (w < 1 || h < 1 || stride < w) {
        set_error("launch_filter: bad geometry");
        return 1;
    }
    if ((uint64_t)w * (uint64_t)h >= (1ull << 29)) {
        set_error("launch_filter: image larger than 2^29 pixels is not supported (seed key packs index in 29 bits)");
        return 1;
    }
    FilterTaps ft;
    if (filter_taps(fc, ft)) return 1;
    const int bands_x = (w + kLaneCols - 1) / kLaneCols;
    const int band_rows = (h + kBandRows - 1) / kBandRows;
    by_begin = std::max(by_begin, 0);
    by_end = std::min(by_end, band_rows);
    if (by_end <= by_begin) return 0;
    const int band_begin = by_begin * bands_x, band_end = by_end * bands_x;
    const int n_wg = ((band_end - band_begin + 3) / 4 + 7) / 8 * 8;  // a multiple of eight: see the XCD mapping in the kernel
    hipLaunchKernelGGL(filter_lanes_kernel, dim3(n_wg), dim3(256), 0, s, img, w, h, stride, ft, dx, dy, dmask,
                       cand, cand_count, tile_max, bands_x, band_begin, band_end);
    LR_HIP(hipGetLastError());
    return 0;
}

int launch_filter(const float* img, int w, int h, int stride, const FilterConsts& fc, float* dx, float* dy,
                  uint8_t* dmask, uint64_t* cand, uint32_t* cand_count, uint32_t* tile_max, hipStream_t s) {
    return launch_filter_rows(img, w, h, stride, fc, dx, dy, dmask, cand, cand_count, tile_max, 0, (h + kBandRows - 1) / kBandRows, s);
}

}  // namespace lramd
