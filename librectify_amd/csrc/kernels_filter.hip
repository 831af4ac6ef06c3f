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

}  // namespace

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
    // the symmetry the kernel relies on is a property of the host's libm results: verify, never assume
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
    dim3 grid(tiles_x(w), tiles_y(h));
    hipLaunchKernelGGL(filter_kernel, grid, dim3(256), 0, s, img, w, h, stride, ft, dx, dy, dmask, cand, cand_count,
                       tile_max);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
