// Does the layout of the planes matter to the flood's gathers?  A wavefront fetches an 8x8 tile of floats, one pixel
// per lane, from pseudo-random tile positions of a 3840x2160 plane:
//   row-major plane: 8 rows x 32 B, each in a different 128-byte line
//   tile-major plane: 256 contiguous bytes (2 lines)
// plus the walk's real pattern (tile + the 36 ring pixels around it, three planes).  Many independent wavefronts,
// DEP dependent fetches each (the walk's chain tile -> neighbour tile).
// RESULT (MI355X): 40 000 wavefronts x 14 fetches: three row-major planes 323 us, row-major float2 + byte 212 us,
// tile-major float2 + byte 143 us; a single chain costs 650-750 ns per fetch whatever the layout.  The flood itself did
// NOT get faster with the tile-major layout (kernels_flood.hip, "What a step costs"): its steps are bound by their own
// bookkeeping, not by these gathers, and the filter's stores got slower.  Kept as the record of that measurement.
// Build: hipcc -O3 --offload-arch=gfx950 -o tile_gather tile_gather.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

constexpr int W = 3840, H = 2160, TX = W / 8, TY = H / 8;

template <int MODE>
__global__ __launch_bounds__(64) void gather(const float* __restrict__ a, const float* __restrict__ b,
                                             const unsigned char* __restrict__ m, int dep, float* out) {
    const int lane = threadIdx.x;
    const int lr = lane >> 3, lc = lane & 7;
    unsigned s = blockIdx.x * 2654435761u + 12345u;
    float acc = 0.f;
    int tx = s % TX, ty = (s / TX) % TY;
    for (int i = 0; i < dep; ++i) {
        float v;
        if (MODE == 0) {  // row-major, one plane
            v = a[(size_t)(ty * 8 + lr) * W + tx * 8 + lc];
        } else if (MODE == 1) {  // tile-major, one plane
            v = a[((size_t)ty * TX + tx) * 64 + lane];
        } else if (MODE == 3) {  // tile + ring, row-major, (dx, dy) interleaved + byte mask
            const float2* ab = reinterpret_cast<const float2*>(a);
            const size_t q = (size_t)(ty * 8 + lr) * W + tx * 8 + lc;
            const float2 p = ab[q];
            v = p.x + p.y + (float)m[q];
            if (lane < 36) {
                const int g = lane >> 3, k = lane & 7;
                const int rx = (g == 0 || g == 1) ? k : (g == 2 ? -1 : (g == 3 ? 8 : ((k & 1) ? 8 : -1)));
                const int ry = (g == 0) ? -1 : (g == 1 ? 8 : ((g == 2 || g == 3) ? k : ((k & 2) ? 8 : -1)));
                const int yy = min(max(ty * 8 + ry, 0), H - 1), xx = min(max(tx * 8 + rx, 0), W - 1);
                const size_t r = (size_t)yy * W + xx;
                const float2 pr = ab[r];
                v += pr.x + pr.y + (float)m[r];
            }
        } else if (MODE == 4) {  // tile + ring, TILE-major, (dx, dy) interleaved + byte mask
            const float2* ab = reinterpret_cast<const float2*>(a);
            const size_t q = ((size_t)ty * TX + tx) * 64 + lane;
            const float2 p = ab[q];
            v = p.x + p.y + (float)m[q];
            if (lane < 36) {
                const int g = lane >> 3, k = lane & 7;
                const int rx = (g == 0 || g == 1) ? k : (g == 2 ? -1 : (g == 3 ? 8 : ((k & 1) ? 8 : -1)));
                const int ry = (g == 0) ? -1 : (g == 1 ? 8 : ((g == 2 || g == 3) ? k : ((k & 2) ? 8 : -1)));
                const int yy = min(max(ty * 8 + ry, 0), H - 1), xx = min(max(tx * 8 + rx, 0), W - 1);
                const size_t r = ((size_t)(yy >> 3) * TX + (xx >> 3)) * 64 + ((yy & 7) << 3 | (xx & 7));
                const float2 pr = ab[r];
                v += pr.x + pr.y + (float)m[r];
            }
        } else if (MODE == 5) {  // tile + ring, row-major, one 16-byte record per pixel
            const float4* rec = reinterpret_cast<const float4*>(a);
            const size_t q = ((size_t)(ty * 8 + lr) * W + tx * 8 + lc) % ((size_t)W * H / 2);
            const float4 p = rec[q];
            v = p.x + p.y + p.z;
            if (lane < 36) {
                const int g = lane >> 3, k = lane & 7;
                const int rx = (g == 0 || g == 1) ? k : (g == 2 ? -1 : (g == 3 ? 8 : ((k & 1) ? 8 : -1)));
                const int ry = (g == 0) ? -1 : (g == 1 ? 8 : ((g == 2 || g == 3) ? k : ((k & 2) ? 8 : -1)));
                const int yy = min(max(ty * 8 + ry, 0), H - 1), xx = min(max(tx * 8 + rx, 0), W - 1);
                const size_t r = ((size_t)yy * W + xx) % ((size_t)W * H / 2);
                const float4 pr = rec[r];
                v += pr.x + pr.y + pr.z;
            }
        } else {  // the walk's pattern: tile + ring, three row-major planes
            const size_t q = (size_t)(ty * 8 + lr) * W + tx * 8 + lc;
            v = a[q] + b[q] + (float)m[q];
            if (lane < 36) {
                const int g = lane >> 3, k = lane & 7;
                const int rx = (g == 0 || g == 1) ? k : (g == 2 ? -1 : (g == 3 ? 8 : ((k & 1) ? 8 : -1)));
                const int ry = (g == 0) ? -1 : (g == 1 ? 8 : ((g == 2 || g == 3) ? k : ((k & 2) ? 8 : -1)));
                const int yy = min(max(ty * 8 + ry, 0), H - 1), xx = min(max(tx * 8 + rx, 0), W - 1);
                const size_t r = (size_t)yy * W + xx;
                v += a[r] + b[r] + (float)m[r];
            }
        }
        acc += v;
        // next tile depends on the data (as the walk's does): a neighbour of the current one
        const unsigned u = __float_as_uint(__shfl(v, 0)) + s;
        s = s * 1664525u + 1013904223u + (u & 1u);
        tx = (tx + 1 + (s >> 16) % 3 - 1 + TX) % TX;
        ty = (ty + (s >> 20) % 3 - 1 + TY) % TY;
    }
    if (acc == 123.456f) out[0] = acc;
}

int main() {
    const size_t n = (size_t)W * H;
    float *a, *b, *o;
    unsigned char* m;
    hipMalloc(&a, n * 8);  // room for the interleaved variants
    hipMalloc(&b, n * 4);
    hipMalloc(&m, n);
    hipMalloc(&o, 4);
    hipMemset(a, 0, n * 8);
    hipMemset(b, 0, n * 4);
    hipMemset(m, 0, n);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves : {256, 4096, 40000}) {
        for (int mode = 0; mode < 6; ++mode) {
            const int dep = waves == 40000 ? 14 : 150;
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(gather<0>, dim3(waves), dim3(64), 0, 0, a, b, m, dep, o);
                if (mode == 1) hipLaunchKernelGGL(gather<1>, dim3(waves), dim3(64), 0, 0, a, b, m, dep, o);
                if (mode == 2) hipLaunchKernelGGL(gather<2>, dim3(waves), dim3(64), 0, 0, a, b, m, dep, o);
                if (mode == 3) hipLaunchKernelGGL(gather<3>, dim3(waves), dim3(64), 0, 0, a, b, m, dep, o);
                if (mode == 4) hipLaunchKernelGGL(gather<4>, dim3(waves), dim3(64), 0, 0, a, b, m, dep, o);
                if (mode == 5) hipLaunchKernelGGL(gather<5>, dim3(waves), dim3(64), 0, 0, a, b, m, dep, o);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (rep) best = ms < best ? ms : best;
            }
            printf("%6d waves x %3d dependent fetches, %s: %8.1f us  (%.0f ns per fetch in a chain)\n", waves, dep,
                   mode == 0 ? "row-major tile                              "
                   : mode == 1 ? "tile-major tile                             "
                   : mode == 2 ? "tile + ring, 3 row-major planes (today)     "
                   : mode == 3 ? "tile + ring, row-major float2 + byte        "
                   : mode == 4 ? "tile + ring, TILE-major float2 + byte       "
                               : "tile + ring, row-major 16-byte records      ",
                   best * 1e3, best * 1e6 / dep);
        }
    }
    return 0;
}
