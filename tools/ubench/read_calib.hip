// FETCH_SIZE calibration for 4-byte-per-lane coalesced reads (the filter kernel's access shape):
// reads N floats once (grid-stride by rows of 64 lanes), so the true HBM read is 4*N bytes.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void read4(const float* __restrict__ p, size_t n, float* out) {
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) s += p[i];
    if (s == 123.456f) out[0] = s;
}
int main() {
    const size_t n = (size_t)512 << 20;  // 2 GiB of floats: far beyond the 256 MiB Infinity Cache
    float *d, *o; hipMalloc(&d, n * 4); hipMalloc(&o, 4); hipMemset(d, 0, n * 4);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, d, n, o);
    hipDeviceSynchronize();
    printf("read4: %zu bytes per launch\n", n * 4);
    return 0;
}
