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
#include "common.h"

namespace lramd {
namespace {

constexpr int IW = 76;  // image tile: cols x0-6 .. x0+69 (16-B aligned windows), rows y0-4 .. y0+35
constexpr int IH = 40;
constexpr int CW = 72;  // conv region: cols x0-4 .. x0+67, rows y0-2 .. y0+33
constexpr int CH = 36;
constexpr int CSTRIPS = CW / 4;

__global__ __launch_bounds__(256) void filter_kernel(const float* __restrict__ img, int w, int h, int stride,
                                                     FilterConsts fc, float* __restrict__ dx_out,
                                                     float* __restrict__ dy_out, uint8_t* __restrict__ dmask_out,
                                                     uint64_t* __restrict__ cand, uint32_t* __restrict__ cand_count,
                                                     uint32_t* __restrict__ tile_max) {
    __shared__ __attribute__((aligned(16))) float s_img[IH][IW];
    __shared__ __attribute__((aligned(16))) float s_mag[CH][CW];
    __shared__ __attribute__((aligned(16))) uint8_t s_bin[CH][CW];
    __shared__ uint32_t s_cnt;
    __shared__ float s_wmax[4];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * kTileW;
    const int y0 = blockIdx.y * kTileH;
    const uint32_t tile = blockIdx.y * gridDim.x + blockIdx.x;

    for (int i = tid; i < IH * IW; i += 256) {
        int r = i / IW, c = i - r * IW;
        int y = y0 - 4 + r, x = x0 - 6 + c;
        float v = 0.f;
        if (y >= 0 && y < h && x >= 0 && x < w) v = img[(size_t)y * stride + x];
        s_img[r][c] = v;
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
                for (int j = 0; j < 5; ++j) {
                    ax = fmaf(win[i][j + p], fc.kx[i * 5 + j], ax);
                    ay = fmaf(win[i][j + p], fc.ky[i * 5 + j], ay);
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
                uint32_t mask = 0;
                if (!row_border && x != 0 && x != w - 1) {  // binary_dilate leaves a 1-px zero border (filter.cpp:52-61)
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) mask |= 1u << s_bin[cr + 1 + i][4 * cs + 3 + p + j];
                }
                dm[p] = (uint8_t)mask;
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
    dim3 grid(tiles_x(w), tiles_y(h));
    hipLaunchKernelGGL(filter_kernel, grid, dim3(256), 0, s, img, w, h, stride, fc, dx, dy, dmask, cand, cand_count,
                       tile_max);
    LR_HIP(hipGetLastError());
    return 0;
}

}  // namespace lramd
