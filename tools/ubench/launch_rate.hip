// How many kernel launches a second does one process get through, from T host threads with a stream each?  (The lanes of a
// batch call enqueue ~60 dependent launches a frame: 13 000 frames/s are 800 000 launches/s.)
// usage: launch_rate [threads] [launches per thread]      build: hipcc --offload-arch=gfx950 -O3 -o launch_rate launch_rate.hip -lpthread
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
__global__ void tiny(unsigned* p) {
    if (threadIdx.x == 0 && blockIdx.x == 0) atomicAdd(p, 1u);
}
int main(int argc, char** argv) {
    const int n = argc > 2 ? std::atoi(argv[2]) : 20000;
    unsigned* d;
    hipMalloc(&d, 4 * 64);
    hipMemset(d, 0, 4 * 64);
    for (int T : {1, 2, 4, 5, 6, 8, 12, 16}) {
        if (argc > 1 && std::atoi(argv[1]) > 0 && T != std::atoi(argv[1])) continue;
        std::vector<hipStream_t> s(T);
        for (auto& x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
        for (int blocks : {1, 1024}) {
            auto t0 = std::chrono::steady_clock::now();
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    for (int i = 0; i < n; ++i) hipLaunchKernelGGL(tiny, dim3(blocks), dim3(64), 0, s[t], d + t);
                    hipStreamSynchronize(s[t]);
                });
            for (auto& x : th) x.join();
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("%2d threads x %d launches of %4d blocks: %.0f launches/s in all, %.2f us per launch per stream\n", T, n, blocks, T * n / dt, dt / n * 1e6);
        }
        for (auto& x : s) hipStreamDestroy(x);
    }
    return 0;
}
