// What does a dependent launch cost on gfx950, and does a HIP graph make it cheaper?  A chain of N kernels that each read a
// word the previous one wrote (like the flood's rounds reading the control block), N = 60: through a stream, through a
// graph captured from the same stream (launched several times), and through a graph of kernel nodes built by hand.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void step(unsigned* w, unsigned grid_note) {
    if (blockIdx.x == 0 && threadIdx.x == 0) w[0] = w[0] + 1u;
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
    unsigned* d;
    CK(hipMalloc(&d, 64));
    CK(hipMemset(d, 0, 64));
    hipStream_t s;
    CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int N = 60;
    for (int grid : {1, 240, 15360}) {
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(step, dim3(grid), dim3(64), 0, s, d, 0u);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("stream, grid %5d: %.2f us per launch\n", grid, ms * 1e3 / N);
        }
        hipGraph_t g;
        hipGraphExec_t ge;
        CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (int i = 0; i < N; ++i) hipLaunchKernelGGL(step, dim3(grid), dim3(64), 0, s, d, 0u);
        CK(hipStreamEndCapture(s, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipEventRecord(e0, s));
            CK(hipGraphLaunch(ge, s));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep) printf("graph,  grid %5d: %.2f us per launch\n", grid, ms * 1e3 / N);
        }
        CK(hipGraphExecDestroy(ge));
        CK(hipGraphDestroy(g));
    }
    unsigned h = 0;
    CK(hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost));
    printf("counter %u\n", h);
    return 0;
}
