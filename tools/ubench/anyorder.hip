// Does hipExtAnyOrderLaunch let two kernels of ONE stream overlap on gfx950?  (hip_ext.h says "not supported on GFX9xx".)
// Two spin kernels of ~200 us each, one workgroup each: back to back they take 400 us, overlapped 200.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
__global__ void spin(unsigned long long ticks, unsigned long long* out) {
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) {}
    if (threadIdx.x == 0) out[blockIdx.x] = wall_clock64() - t0;
}
int main() {
    unsigned long long* d;
    hipMalloc(&d, 64);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const unsigned long long ticks = 20000;  // 100 MHz clock: 200 us
    for (int mode = 0; mode < 2; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0, s);
            hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d);
            if (mode == 0) hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, ticks, d + 1);
            else hipExtLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, nullptr, nullptr, hipExtAnyOrderLaunch, ticks, d + 1);
            hipEventRecord(e1, s);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s: %.1f us for two 200 us kernels\n", mode ? "second launch any-order" : "plain launches", ms * 1e3);
        }
    return 0;
}
