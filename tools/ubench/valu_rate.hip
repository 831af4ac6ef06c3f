// Micro-benchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 on gfx950 (independent chains, all CUs busy).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float float2v __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, float a, float b, int iters) {
    float x[8];
    float2v y[8];
    for (int i = 0; i < 8; ++i) { x[i] = threadIdx.x * 0.001f + i; y[i] = float2v{x[i], x[i] + 1.f}; }
    float2v a2{a, a}, b2{b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) x[i] = __builtin_fmaf(x[i], a, b);
            else y[i] = __builtin_elementwise_fma(y[i], a2, b2);
        }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += (MODE == 0) ? x[i] : (y[i].x + y[i].y);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float* d; hipMalloc(&d, 256 * 4096 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 4096, blocks = 256 * 8;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            else hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double inst = (double)blocks * 4 /*waves*/ * iters * 8;
            double flops = inst * 64 * 2 * (mode ? 2 : 1);
            printf("%s: %.3f ms, %.2f G wave-instr/s, %.1f TFLOP/s, %.2f cycles/instr/SIMD @2.4GHz\n", mode ? "v_pk_fma_f32" : "v_fma_f32", ms,
                   inst / ms / 1e6, flops / ms / 1e9, 1024.0 * 2.4e9 / (inst / (ms * 1e-3)));
        }
    }
    return 0;
}
