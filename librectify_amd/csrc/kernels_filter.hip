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
//   write per-tile max(mag)         4 B/tile
//
// A 256-thread workgroup owns a 64x32 output tile.  The image tile with a 4-px halo
// (conv radius 2 + NMS radius 2) is staged in LDS; the 5x5 correlation is evaluated on the
// tile grown by 2 px (so mag/bin of the NMS/dilate halo are recomputed, not exchanged), each
// thread producing 1x4 strips from an aligned 5x8 LDS window.  Arithmetic is the canonical
// form shared with the CPU oracle: acc = fmaf(img, K, acc) in row-major tap order,
// mag = sqrtf(dx*dx + dy*dy) without contraction, bin = first strict argmax of
// |fmaf(dx, sin, dy*cos)|.
#include <cstdlib>
#include <cstring>

#include "common.h"

namespace lramd {
namespace {

constexpr int IW = 76;  // image tile: cols x0-6 .. x0+69 (16-B aligned windows), rows y0-4 .. y0+35
constexpr int IH = 40;
constexpr int CW = 72;  // conv region: cols x0-4 .. x0+67, rows y0-2 .. y0+33
constexpr int CH = 36;
constexpr int CSTRIPS = CW / 4;

// The 5x5 Gaussian-derivative taps as the host computes them (filter.cpp:72-75) satisfy, bit for bit,
//   Hx[i][j] = -Hx[i][4-j] = Hx[4-i][j],  Hx[i][2] = +0      Hy[i][j] = Hx[j][i]
// (the sign enters only through z, and exp() sees the same argument), so six magnitudes describe
// both kernels.  The kernel keeps them in SGPRs; a negated tap is an FMA source modifier, and the
// zero taps are skipped: fmaf(v, +0, acc) == acc for every finite v because acc is never -0
// (it starts at +0 and x + (-x) rounds to +0).
struct FilterTaps {
    float k[3][2];    // Hx[i][j] for i = 0..2 (|y| = 2,1,0), j = 0..1 (x = -2,-1)
    float st[kBins];
    float ct[kBins];
};

__global__ __launch_bounds__(256) void filter_kernel(const float* __restrict__ img, int w, int h, int stride,
                                                     FilterTaps fc, float* __restrict__ dx_out,
                                                     float* __restrict__ dy_out, uint8_t* __restrict__ dmask_out,
                                                     uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_count,
                                                     uint32_t* __restrict__ tile_max) {
    __shared__ __attribute__((aligned(16))) float s_img[IH][IW];
    __shared__ __attribute__((aligned(16))) float s_mag[CH][CW];
    __shared__ __attribute__((aligned(16))) uint8_t s_bin[CH][CW];       // bin index (for the peak records)
    __shared__ __attribute__((aligned(16))) uint8_t s_bit[CH][CW + 8];   // 1 << bin, stored one column to the right
    __shared__ uint32_t s_cnt;
    __shared__ float s_wmax[4];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kTileW;
    const int y0 = blockIdx.y * kTileH;
    const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;

    // image tile: 20 aligned float4 groups per row (x0-8 .. x0+71); the LDS copy starts at x0-6 so
    // that the 5x8 conv windows are 16-B aligned, hence each group lands as two 8-B halves
    const bool in_vec = ((stride & 3) == 0) && ((reinterpret_cast<uintptr_t>(img) & 15) == 0);
    for (int i = tid; i < IH * 20; i += 256) {
        const int r = i / 20, m = i - r * 20;
        const int y = y0 - 4 + r, x = x0 - 8 + 4 * m;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (y >= 0 && y < h) {
            const float* row = img + (size_t)y * stride;
            if (in_vec && x >= 0 && x + 3 < w) {
                v = *reinterpret_cast<const float4*>(row + x);
            } else {
                if (x >= 0 && x < w) v.x = row[x];
                if (x + 1 >= 0 && x + 1 < w) v.y = row[x + 1];
                if (x + 2 >= 0 && x + 2 < w) v.z = row[x + 2];
                if (x + 3 >= 0 && x + 3 < w) v.w = row[x + 3];
            }
        }
        const int c = 4 * m - 2;
        if (m > 0) *reinterpret_cast<float2*>(&s_img[r][c]) = make_float2(v.x, v.y);
        if (m < 19) *reinterpret_cast<float2*>(&s_img[r][c + 2]) = make_float2(v.z, v.w);
    }
    if (tid == 0) s_cnt = 0;
    __syncthreads();

    const bool vec_ok = (w & 3) == 0;

    for (int sidx = tid; sidx < CH * CSTRIPS; sidx += 256) {
        const int sr = sidx / CSTRIPS, sc = sidx - sr * CSTRIPS;
        float win[5][8];
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const float4 a = *reinterpret_cast<const float4*>(&s_img[sr + i][4 * sc]);
            const float4 b = *reinterpret_cast<const float4*>(&s_img[sr + i][4 * sc + 4]);
            win[i][0] = a.x; win[i][1] = a.y; win[i][2] = a.z; win[i][3] = a.w;
            win[i][4] = b.x; win[i][5] = b.y; win[i][6] = b.z; win[i][7] = b.w;
        }
        float ddx[4], ddy[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float ax = 0.f, ay = 0.f;
#pragma unroll
            for (int i = 0; i < 5; ++i)
#pragma unroll
                for (int j = 0; j < 5; ++j) {  // row-major tap order, as the oracle
                    const int ii = i < 3 ? i : 4 - i, jj = j < 3 ? j : 4 - j;
                    if (j != 2) {
                        const float kx = fc.k[ii][jj];  // |Hx[i][j]|, sign by column
                        ax = (j < 2) ? fmaf(win[i][j + p], kx, ax) : fmaf(win[i][j + p], -kx, ax);
                    }
                    if (i != 2) {
                        const float ky = fc.k[jj][ii];  // Hy = Hx^T
                        ay = (i < 2) ? fmaf(win[i][j + p], ky, ay) : fmaf(win[i][j + p], -ky, ay);
                    }
                }
            ddx[p] = ax;
            ddy[p] = ay;
        }
        const int y = y0 - 2 + sr;
        const int xb = x0 - 4 + 4 * sc;
        const bool row_ok = (y >= 2) && (y < h - 2);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = xb + p;
            const bool ok = row_ok && (x >= 2) && (x < w - 2);  // conv_2d leaves a zero border (filter.cpp:89-97)
            const float vx = ok ? ddx[p] : 0.f;
            const float vy = ok ? ddy[p] : 0.f;
            ddx[p] = vx;
            ddy[p] = vy;
            const float m = sqrtf(vx * vx + vy * vy);
            int bin = 0;  // grad_bin is left uninitialised by the reference where all planes are 0; canonical 0
            float gmax = 0.f;
#pragma unroll
            for (int b = 0; b < kBins; ++b) {
                const float g = directional(vx, vy, fc.st[b], fc.ct[b]);
                if (g > gmax) {
                    bin = b;
                    gmax = g;
                }
            }
            s_mag[sr][4 * sc + p] = m;
            s_bin[sr][4 * sc + p] = (uint8_t)bin;
            s_bit[sr][4 * sc + p + 1] = (uint8_t)(1u << bin);
        }
        // core strips write dx, dy straight from registers
        if (sr >= 2 && sr < CH - 2 && sc >= 1 && sc <= 16 && y < h && xb < w) {
            const size_t o = (size_t)y * w + xb;
            if (vec_ok && xb + 3 < w) {
                *reinterpret_cast<float4*>(dx_out + o) = make_float4(ddx[0], ddx[1], ddx[2], ddx[3]);
                *reinterpret_cast<float4*>(dy_out + o) = make_float4(ddy[0], ddy[1], ddy[2], ddy[3]);
            } else {
#pragma unroll
                for (int p = 0; p < 4; ++p)
                    if (xb + p < w) {
                        dx_out[o + p] = ddx[p];
                        dy_out[o + p] = ddy[p];
                    }
            }
        }
    }
    __syncthreads();

    float lmax = 0.f;
    for (int sidx = tid; sidx < kTileH * (kTileW / 4); sidx += 256) {
        const int cr = sidx / (kTileW / 4), cs = sidx - cr * (kTileW / 4);
        const int y = y0 + cr, xb = x0 + 4 * cs;
        if (y >= h || xb >= w) continue;
        // 5-row column maxima over conv-region cols 4cs+2 .. 4cs+9 (rows cr .. cr+4)
        float colmax[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float m = s_mag[cr][4 * cs + 2 + j];
#pragma unroll
            for (int i = 1; i < 5; ++i) m = fmaxf(m, s_mag[cr + i][4 * cs + 2 + j]);
            colmax[j] = m;
        }
        const bool row_in = (y >= 2) && (y < h - 2);
        const bool row_border = (y == 0) || (y == h - 1);
        // dilated-bin mask of the 4 pixels at once: the 3x3 windows span conv cols 4cs+3 .. 4cs+8, i.e. the
        // two aligned words at s_bit cols 4cs+4 and 4cs+8 of rows cr+1 .. cr+3; byte p of
        // (W | W>>8 | W>>16) is the OR of bytes p..p+2
        uint32_t dm4 = 0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t lo = *reinterpret_cast<const uint32_t*>(&s_bit[cr + 1 + i][4 * cs + 4]);
            const uint32_t hi = *reinterpret_cast<const uint32_t*>(&s_bit[cr + 1 + i][4 * cs + 8]);
            const uint64_t W = ((uint64_t)hi << 32) | lo;
            dm4 |= (uint32_t)(W | (W >> 8) | (W >> 16));
        }
        uint8_t dm[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int x = xb + p;
            dm[p] = 0;
            if (x < w) {
                const float center = s_mag[cr + 2][4 * cs + 4 + p];
                lmax = fmaxf(lmax, center);
                float mx = colmax[p];
#pragma unroll
                for (int j = 1; j < 5; ++j) mx = fmaxf(mx, colmax[p + j]);
                // binary_dilate leaves a 1-px zero border (filter.cpp:52-61)
                dm[p] = (!row_border && x != 0 && x != w - 1) ? (uint8_t)(dm4 >> (8 * p)) : (uint8_t)0;
                const bool peak = row_in && (x >= 2) && (x < w - 2) && (center > 0.f) && (center == mx);
                if (peak) {
                    const uint32_t slot = atomicAdd(&s_cnt, 1u);
                    const uint32_t idx = (uint32_t)y * (uint32_t)w + (uint32_t)x;
                    const uint32_t bin = s_bin[cr + 2][4 * cs + 4 + p];
                    cand[(size_t)tile * kCandPerTile + slot] =
                        ((uint64_t)__float_as_uint(center) << 32) | (uint64_t)((idx << 3) | bin);
                }
            }
        }
        const size_t o = (size_t)y * w + xb;
        if (vec_ok && xb + 3 < w) {
            *reinterpret_cast<uint32_t*>(dmask_out + o) =
                (uint32_t)dm[0] | ((uint32_t)dm[1] << 8) | ((uint32_t)dm[2] << 16) | ((uint32_t)dm[3] << 24);
        } else {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (xb + p < w) dmask_out[o + p] = dm[p];
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if ((tid & 63) == 0) s_wmax[tid >> 6] = lmax;
    __syncthreads();
    if (tid == 0) {
        const float m = fmaxf(fmaxf(s_wmax[0], s_wmax[1]), fmaxf(s_wmax[2], s_wmax[3]));
        tile_max[tile] = __float_as_uint(m);
        cand_count[tile] = s_cnt;
    }
}


// ---------------------------------------------------------------------------------------------
// Row-streaming variant (default).  One wavefront owns a band of 120 columns x 32 rows and walks
// it top to bottom, two adjacent pixels per lane (lanes 0-1 and 62-63 are halo).  Everything
// rolls in registers: the last five image rows (own float2 plus the neighbours' via lane shifts),
// five rows of magnitudes and of direction bits.  No LDS, no barriers, no per-pixel index math;
// every global access is a coalesced row segment (512 B loads, 512 B + 512 B + 128 B stores), and
// the 5x5 correlation is computed once per pixel except for the band halo (1.2x instead of the
// tile kernel's 1.5x).  Same canonical arithmetic, same outputs as filter_kernel above.
constexpr int kBandCols = 120;
constexpr int kBandRows = 30;  // 2160 = 72 x 30: no ragged last band at 4K, 4968 waves = 4.85 per SIMD
constexpr int kBandSteps = kBandRows + 8;  // image rows y0-4 .. y0+35

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

struct Roll {
    float win[5][6];   // image rows t-4..t (slot = row mod 5), cols xc-2 .. xc+3
    float mag[5][2];   // own magnitudes of the last five conv rows
    uint32_t bits[5];  // (1 << bin) of the two own pixels, packed b0 | b1 << 8
    float2 q[5];       // prefetched image rows
};

__device__ __forceinline__ float2 load_row2(const float* __restrict__ img, int stride, int w, int h, int yr, int xc,
                                            bool vec2) {
    float2 v = make_float2(0.f, 0.f);
    if (yr >= 0 && yr < h) {
        const float* row = img + (size_t)yr * stride;
        if (vec2 && xc >= 0 && xc + 1 < w) {
            v = *reinterpret_cast<const float2*>(row + xc);
        } else {
            if (xc >= 0 && xc < w) v.x = row[xc];
            if (xc + 1 >= 0 && xc + 1 < w) v.y = row[xc + 1];
        }
    }
    return v;
}

template <int K>
__device__ __forceinline__ void band_step(Roll& R, const int t, const float* __restrict__ img, const int w, const int h,
                                          const int stride, const FilterTaps& fc, float* __restrict__ dx_out,
                                          float* __restrict__ dy_out, uint8_t* __restrict__ dmask_out,
                                          uint64_t* __restrict__ cand_band, uint32_t& ncand, float& lmax, const int y0,
                                          const int xc, const int lane, const bool vec2, const bool useful) {
    // newest image row -> slot K; prefetch the row four steps ahead into the slot it will be read from
    const float2 cur = R.q[K];
    R.q[(K + 4) % 5] = load_row2(img, stride, w, h, y0 - 4 + t + 4, xc, vec2);
    R.win[K][0] = from_lower(cur.x);
    R.win[K][1] = from_lower(cur.y);
    R.win[K][2] = cur.x;
    R.win[K][3] = cur.y;
    R.win[K][4] = from_upper(cur.x);
    R.win[K][5] = from_upper(cur.y);
    if (t < 4) return;  // wave-uniform: the window is not full yet

    // ---- conv row yc = yr - 2 -----------------------------------------------------------------
    const int yc = y0 - 6 + t;
    float ddx[2], ddy[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        float ax = 0.f, ay = 0.f;
#pragma unroll
        for (int i = 0; i < 5; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) {  // row-major tap order, as the oracle
                const float v = R.win[(K + 1 + i) % 5][p + j];
                const int ii = i < 3 ? i : 4 - i, jj = j < 3 ? j : 4 - j;
                if (j != 2) ax = (j < 2) ? fmaf(v, fc.k[ii][jj], ax) : fmaf(v, -fc.k[ii][jj], ax);
                if (i != 2) ay = (i < 2) ? fmaf(v, fc.k[jj][ii], ay) : fmaf(v, -fc.k[jj][ii], ay);
            }
        ddx[p] = ax;
        ddy[p] = ay;
    }
    const bool row_ok = (yc >= 2) && (yc < h - 2);
    uint32_t bits = 0;
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const int x = xc + p;
        const bool ok = row_ok && (x >= 2) && (x < w - 2);  // conv_2d leaves a zero border (filter.cpp:89-97)
        const float vx = ok ? ddx[p] : 0.f;
        const float vy = ok ? ddy[p] : 0.f;
        ddx[p] = vx;
        ddy[p] = vy;
        R.mag[K][p] = sqrtf(vx * vx + vy * vy);
        int bin = 0;
        float gmax = 0.f;
#pragma unroll
        for (int b = 0; b < kBins; ++b) {
            const float g = directional(vx, vy, fc.st[b], fc.ct[b]);
            bin = (g > gmax) ? b : bin;
            gmax = fmaxf(gmax, g);
        }
        bits |= (1u << bin) << (8 * p);
    }
    R.bits[K] = bits;
    if (useful && yc >= y0 && yc < y0 + kBandRows && yc < h && xc < w) {
        const size_t o = (size_t)yc * w + xc;
        if (((w & 1) == 0) && xc + 1 < w) {
            *reinterpret_cast<float2*>(dx_out + o) = make_float2(ddx[0], ddx[1]);
            *reinterpret_cast<float2*>(dy_out + o) = make_float2(ddy[0], ddy[1]);
        } else {
            dx_out[o] = ddx[0];
            dy_out[o] = ddy[0];
            if (xc + 1 < w) {
                dx_out[o + 1] = ddx[1];
                dy_out[o + 1] = ddy[1];
            }
        }
    }
    if (t < 8) return;  // wave-uniform

    // ---- NMS + dilated mask for output row yo = yc - 2 -------------------------------------------
    const int yo = yc - 2;
    if (yo >= h) return;  // wave-uniform
    const float cm0 = fmaxf(fmaxf(fmaxf(R.mag[0][0], R.mag[1][0]), fmaxf(R.mag[2][0], R.mag[3][0])), R.mag[4][0]);
    const float cm1 = fmaxf(fmaxf(fmaxf(R.mag[0][1], R.mag[1][1]), fmaxf(R.mag[2][1], R.mag[3][1])), R.mag[4][1]);
    const float l0 = from_lower(cm0), l1 = from_lower(cm1);
    const float r0 = from_upper(cm0), r1 = from_upper(cm1);
    const float mx0 = fmaxf(fmaxf(fmaxf(l0, l1), fmaxf(cm0, cm1)), r0);
    const float mx1 = fmaxf(fmaxf(fmaxf(l1, cm0), fmaxf(cm1, r0)), r1);
    const float c0 = R.mag[(K + 3) % 5][0], c1 = R.mag[(K + 3) % 5][1];
    const uint32_t cb = R.bits[(K + 3) % 5];
    const uint32_t v = R.bits[(K + 2) % 5] | cb | R.bits[(K + 4) % 5];
    const uint32_t lv = from_lower(v), rv = from_upper(v);
    uint32_t dm0 = ((lv >> 8) | v | (v >> 8)) & 0xFFu;
    uint32_t dm1 = (v | (v >> 8) | rv) & 0xFFu;
    const bool row_in = (yo >= 2) && (yo < h - 2);
    const bool row_border = (yo == 0) || (yo == h - 1);
    const bool in0 = useful && xc < w, in1 = useful && xc + 1 < w;
    if (row_border || xc == 0 || xc == w - 1) dm0 = 0;  // binary_dilate leaves a 1-px zero border (filter.cpp:52-61)
    if (row_border || xc + 1 == w - 1) dm1 = 0;
    if (in0) {
        lmax = fmaxf(lmax, c0);
        const size_t o = (size_t)yo * w + xc;
        if (((w & 1) == 0) && in1) {
            *reinterpret_cast<uint16_t*>(dmask_out + o) = (uint16_t)(dm0 | (dm1 << 8));
        } else {
            dmask_out[o] = (uint8_t)dm0;
            if (in1) dmask_out[o + 1] = (uint8_t)dm1;
        }
    }
    if (in1) lmax = fmaxf(lmax, c1);
    const bool peak0 = in0 && row_in && (xc >= 2) && (xc < w - 2) && (c0 > 0.f) && (c0 == mx0);
    const bool peak1 = in1 && row_in && (xc + 1 >= 2) && (xc + 1 < w - 2) && (c1 > 0.f) && (c1 == mx1);
    const uint64_t m0 = __ballot(peak0), m1 = __ballot(peak1);
    if (m0 | m1) {  // wave-uniform, rare
        const uint64_t below = (1ull << lane) - 1ull;
        if (peak0) {
            const uint32_t idx = (uint32_t)yo * (uint32_t)w + (uint32_t)xc;
            const uint32_t bin = (uint32_t)__ffs((int)(cb & 0xFFu)) - 1u;
            cand_band[ncand + (uint32_t)__popcll(m0 & below)] =
                ((uint64_t)__float_as_uint(c0) << 32) | (uint64_t)((idx << 3) | bin);
        }
        ncand += (uint32_t)__popcll(m0);
        if (peak1) {
            const uint32_t idx = (uint32_t)yo * (uint32_t)w + (uint32_t)xc + 1u;
            const uint32_t bin = (uint32_t)__ffs((int)((cb >> 8) & 0xFFu)) - 1u;
            cand_band[ncand + (uint32_t)__popcll(m1 & below)] =
                ((uint64_t)__float_as_uint(c1) << 32) | (uint64_t)((idx << 3) | bin);
        }
        ncand += (uint32_t)__popcll(m1);
    }
}

__global__ __launch_bounds__(256) void filter_rows_kernel(const float* __restrict__ img, int w, int h, int stride,
                                                          FilterTaps fc, float* __restrict__ dx_out,
                                                          float* __restrict__ dy_out, uint8_t* __restrict__ dmask_out,
                                                          uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_count,
                                                          uint32_t* __restrict__ tile_max, int bands_x, int n_bands) {
    const int lane = threadIdx.x & 63;
    const int band = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (band >= n_bands) return;
    const int by = band / bands_x, bx = band - by * bands_x;
    const int y0 = by * kBandRows;
    const int xc = bx * kBandCols - 4 + 2 * lane;
    const bool vec2 = ((stride & 1) == 0) && ((reinterpret_cast<uintptr_t>(img) & 7) == 0);
    const bool useful = lane >= 2 && lane <= 61;
    Roll R;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < 6; ++j) R.win[i][j] = 0.f;
        R.mag[i][0] = R.mag[i][1] = 0.f;
        R.bits[i] = 0x0101u;
        R.q[i] = make_float2(0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) R.q[i] = load_row2(img, stride, w, h, y0 - 4 + i, xc, vec2);
    uint64_t* cand_band = cand + (size_t)band * (kBandCols * kBandRows);
    uint32_t ncand = 0;
    float lmax = 0.f;
    for (int t0 = 0; t0 < kBandSteps; t0 += 5) {
        band_step<0>(R, t0 + 0, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, xc, lane, vec2, useful);
        band_step<1>(R, t0 + 1, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, xc, lane, vec2, useful);
        band_step<2>(R, t0 + 2, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, xc, lane, vec2, useful);
        band_step<3>(R, t0 + 3, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, xc, lane, vec2, useful);
        band_step<4>(R, t0 + 4, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, xc, lane, vec2, useful);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) lmax = fmaxf(lmax, __shfl_xor(lmax, off));
    if (lane == 0) {
        tile_max[band] = __float_as_uint(lmax);
        cand_count[band] = ncand;
    }
}

// ---------------------------------------------------------------------------------------------
// Same row-streaming scheme with ONE pixel per lane: a wavefront owns 56 columns (lanes 4..59; four
// halo lanes on each side) x 32 rows.  Half the serial instruction stream per wavefront and twice
// the wavefronts of the two-pixel variant: the per-wave VALU stream, not bandwidth, is what bounds
// this kernel, so more, shorter waves win.
constexpr int kLaneCols = 56;

struct Roll1 {
    float win[5][5];   // image rows t-4..t (slot = row mod 5), cols x-2 .. x+2
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

// INTERIOR: every column of the 64 lanes and every row the band touches is at least 2 px inside the image, so all
// border, clamp and zero-border tests are compile-time true (the common case: all but the outer ring of bands).
template <int K10, bool INTERIOR>
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
    const float l1 = from_lower(cur), r1 = from_upper(cur);
    R.win[K][0] = from_lower(l1);
    R.win[K][1] = l1;
    R.win[K][2] = cur;
    R.win[K][3] = r1;
    R.win[K][4] = from_upper(r1);
    if (t < 4) return;  // wave-uniform: the window is not full yet

    const int yc = y0 - 6 + t;  // conv row
    float ax = 0.f, ay = 0.f;
#pragma unroll
    for (int i = 0; i < 5; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) {  // row-major tap order, as the oracle
            const float v = R.win[(K + 1 + i) % 5][j];
            const int ii = i < 3 ? i : 4 - i, jj = j < 3 ? j : 4 - j;
            if (j != 2) ax = (j < 2) ? fmaf(v, fc.k[ii][jj], ax) : fmaf(v, -fc.k[ii][jj], ax);
            if (i != 2) ay = (i < 2) ? fmaf(v, fc.k[jj][ii], ay) : fmaf(v, -fc.k[jj][ii], ay);
        }
    const bool ok = INTERIOR || ((yc >= 2) && (yc < h - 2) && (x >= 2) && (x < w - 2));  // conv_2d's zero border (filter.cpp:89-97)
    const float vx = ok ? ax : 0.f;
    const float vy = ok ? ay : 0.f;
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
    const bool in_img = useful && (INTERIOR || x < w);
    if (t >= 6 && t < 6 + kBandRows && (INTERIOR || yc < h)) {  // yc in [y0, y0+32): wave-uniform, compile-time in t
        if (in_img) {
            const unsigned row = (unsigned)yc * (unsigned)w * 4u;
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vx), r_dx, (unsigned)x * 4u, row, 0);
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, vy), r_dy, (unsigned)x * 4u, row, 0);
        }
    }
    if (t < 8) return;  // wave-uniform

    const int yo = yc - 2;  // output row of the NMS / dilation
    if (!INTERIOR && yo >= h) return;  // wave-uniform
    const float cm = fmaxf(fmaxf(fmaxf(R.mag[0], R.mag[1]), fmaxf(R.mag[2], R.mag[3])), R.mag[4]);
    const float cl1 = from_lower(cm), cr1 = from_upper(cm);
    const float cl2 = from_lower(cl1), cr2 = from_upper(cr1);
    const float mx = fmaxf(fmaxf(fmaxf(cl2, cl1), fmaxf(cm, cr1)), cr2);
    const float c = R.mag[(K + 3) % 5];
    const uint32_t cb = R.bits[(K + 3) % 5];
    const uint32_t v = R.bits[(K + 2) % 5] | cb | R.bits[(K + 4) % 5];
    uint32_t dm = from_lower(v) | v | from_upper(v);
    if (!INTERIOR && (yo == 0 || yo == h - 1 || x == 0 || x == w - 1)) dm = 0;  // binary_dilate's 1-px zero border (filter.cpp:52-61)
    if (in_img) {
        lmax = fmaxf(lmax, c);
        __builtin_amdgcn_raw_buffer_store_b8((unsigned char)dm, r_dm, (unsigned)x, (unsigned)yo * (unsigned)w, 0);
    }
    const bool peak = in_img && (INTERIOR || ((yo >= 2) && (yo < h - 2) && (x >= 2) && (x < w - 2))) && (c > 0.f) && (c == mx);
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
                                                           uint32_t* __restrict__ tile_max, int bands_x, int n_bands) {
    const int lane = threadIdx.x & 63;
    // the band index is the same in all 64 lanes: say so, and every row pointer, row test and loop bound below
    // lives in SGPRs (scalar base + 32-bit lane offset addressing) instead of 64-bit VGPR arithmetic
    const int band = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (band >= n_bands) return;
    const int by = band / bands_x, bx = band - by * bands_x;
    const int y0 = by * kBandRows;
    const int x = bx * kLaneCols - 4 + lane;
    const int xcl = min(max(x, 0), w - 1);
    const bool useful = lane >= 4 && lane <= 59;
    Roll1 R;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
#pragma unroll
        for (int j = 0; j < 5; ++j) R.win[i][j] = 0.f;
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
#define LR_STEP(k, I) lane_step<k, I>(R, t0 + k, img, w, h, stride, fc, dx_out, dy_out, dmask_out, cand_band, ncand, lmax, y0, x, xcl, lane, useful, r_img, r_dx, r_dy, r_dm)
    // fully unrolled: across a loop back-edge the compiler can only wait for vmcnt(0), which would expose the
    // latency of every store in flight once per iteration.  (An INTERIOR = true instantiation for bands away
    // from the image border was measured: no gain, twice the code; the kernel is bound by VALU issue.)
#pragma unroll
    for (int t0 = 0; t0 < (kBandSteps + 9) / 10 * 10; t0 += 10) {
        LR_STEP(0, false); LR_STEP(1, false); LR_STEP(2, false); LR_STEP(3, false); LR_STEP(4, false);
        LR_STEP(5, false); LR_STEP(6, false); LR_STEP(7, false); LR_STEP(8, false); LR_STEP(9, false);
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

// 0 = 64x32 LDS tiles, 1 = row streaming with 2 px/lane, 2 = row streaming with 1 px/lane (default)
static int filter_variant() {
    static const int v = [] {
        const char* e = std::getenv("LIBRECTIFY_FILTER");
        if (e && std::strcmp(e, "tile") == 0) return 0;
        if (e && std::strcmp(e, "rows2") == 0) return 1;
        return 2;
    }();
    return v;
}

FilterGeom filter_geometry(int w, int h) {
    FilterGeom g;
    if (filter_variant() == 2) {
        g.n_tiles = ((w + kLaneCols - 1) / kLaneCols) * ((h + kBandRows - 1) / kBandRows);
        g.cand_cap = kLaneCols * kBandRows;
    } else if (filter_variant() == 1) {
        g.n_tiles = ((w + kBandCols - 1) / kBandCols) * ((h + kBandRows - 1) / kBandRows);
        g.cand_cap = kBandCols * kBandRows;
    } else {
        g.n_tiles = tiles_x(w) * tiles_y(h);
        g.cand_cap = kCandPerTile;
    }
    return g;
}

int launch_filter(const float* img, int w, int h, int stride, const FilterConsts& fc, float* dx, float* dy,
                  uint8_t* dmask, uint64_t* cand, uint32_t* cand_count, uint32_t* tile_max, hipStream_t s) {
    if (w < 1 || h < 1 || stride < w) {
        set_error("launch_filter: bad geometry");
        return 1;
    }
    if ((uint64_t)w * (uint64_t)h >= (1ull << 29)) {
        set_error("launch_filter: image larger than 2^29 pixels is not supported (seed key packs index in 29 bits)");
        return 1;
    }
    FilterTaps ft;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) ft.k[i][j] = fc.kx[i * 5 + j];
    // the symmetry the kernels rely on is a property of the host's libm results: verify, never assume
    for (int i = 0; i < 5; ++i)
        for (int j = 0; j < 5; ++j) {
            const int ii = i < 3 ? i : 4 - i, jj = j < 3 ? j : 4 - j;
            const float ex = (j == 2) ? 0.f : (j < 2 ? ft.k[ii][jj] : -ft.k[ii][jj]);
            const float ey = (i == 2) ? 0.f : (i < 2 ? ft.k[jj][ii] : -ft.k[jj][ii]);
            if (!(ex == fc.kx[i * 5 + j]) || !(ey == fc.ky[i * 5 + j])) {
                set_error("launch_filter: derivative taps are not (anti)symmetric on this host");
                return 1;
            }
        }
    for (int b = 0; b < kBins; ++b) {
        ft.st[b] = fc.st[b];
        ft.ct[b] = fc.ct[b];
    }
    if (filter_variant() == 2) {
        const int bands_x = (w + kLaneCols - 1) / kLaneCols;
        const int n_bands = bands_x * ((h + kBandRows - 1) / kBandRows);
        hipLaunchKernelGGL(filter_lanes_kernel, dim3((n_bands + 3) / 4), dim3(256), 0, s, img, w, h, stride, ft, dx, dy,
                           dmask, cand, cand_count, tile_max, bands_x, n_bands);
    } else if (filter_variant() == 1) {
        const int bands_x = (w + kBandCols - 1) / kBandCols;
        const int n_bands = bands_x * ((h + kBandRows - 1) / kBandRows);
        hipLaunchKernelGGL(filter_rows_kernel, dim3((n_bands + 3) / 4), dim3(256), 0, s, img, w, h, stride, ft, dx, dy,
                           dmask, cand, cand_count, tile_max, bands_x, n_bands);
    } else {
        dim3 grid(tiles_x(w), tiles_y(h));
        hipLaunchKernelGGL(filter_kernel, grid, dim3(256), 0, s, img, w, h, stride, ft, dx, dy, dmask, cand, cand_count,
                           tile_max);
    }
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
