// How should a 4K frame (33 MB fp32) get from a caller's pageable buffer into HBM?  Times the candidates:
//   a) hipMemcpy from pageable memory (the runtime stages it internally)
//   b) memcpy into a pinned staging buffer with T host threads, then one DMA
//   c) the same, pipelined in row bands (copy of band k+1 overlaps the DMA of band k)
//   d) hipHostRegister + DMA + hipHostUnregister
//   e) DMA from memory that is already pinned (the floor)
// Build: hipcc -O2 --offload-arch=gfx950 -o h2d_paths h2d_paths.hip -lpthread
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_memcpy(char* dst, const char* src, size_t bytes, int T) {
    if (T <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> th;
    const size_t per = (bytes / T + 4095) & ~(size_t)4095;
    for (int t = 0; t < T; ++t) {
        const size_t b = (size_t)t * per, e = std::min(bytes, b + per);
        if (b < e) th.emplace_back([=] { std::memcpy(dst + b, src + b, e - b); });
    }
    for (auto& x : th) x.join();
}

int main() {
    const size_t bytes = (size_t)3840 * 2160 * 4;
    const int reps = 10;
    std::vector<char*> page(reps);
    for (auto& p : page) {
        p = (char*)std::malloc(bytes);
        std::memset(p, 1, bytes);  // touch: the pages exist
    }
    char *pin, *dev;
    hipHostMalloc((void**)&pin, bytes);
    hipMalloc((void**)&dev, bytes);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice);

    double t = now();
    for (int r = 0; r < reps; ++r) hipMemcpy(dev, page[r], bytes, hipMemcpyHostToDevice);
    printf("a) hipMemcpy from pageable:            %.3f ms\n", (now() - t) / reps * 1e3);

    for (int T : {1, 2, 4, 8}) {
        t = now();
        double tc = 0;
        for (int r = 0; r < reps; ++r) {
            const double t0 = now();
            par_memcpy(pin, page[r], bytes, T);
            tc += now() - t0;
            hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s);
            hipStreamSynchronize(s);
        }
        printf("b) memcpy to pinned (%d thr) + DMA:      %.3f ms (memcpy alone %.3f)\n", T, (now() - t) / reps * 1e3, tc / reps * 1e3);
    }
    for (int T : {1, 4}) {
        const int bands = 8;
        t = now();
        for (int r = 0; r < reps; ++r) {
            const size_t per = bytes / bands;
            for (int k = 0; k < bands; ++k) {
                par_memcpy(pin + k * per, page[r] + k * per, per, T);
                hipMemcpyAsync(dev + k * per, pin + k * per, per, hipMemcpyHostToDevice, s);
            }
            hipStreamSynchronize(s);
        }
        printf("c) banded (8) memcpy (%d thr) + DMA:     %.3f ms\n", T, (now() - t) / reps * 1e3);
    }
    t = now();
    double treg = 0;
    for (int r = 0; r < reps; ++r) {
        const double t0 = now();
        if (hipHostRegister(page[r], bytes, hipHostRegisterDefault) != hipSuccess) printf("register failed\n");
        treg += now() - t0;
        hipMemcpyAsync(dev, page[r], bytes, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        hipHostUnregister(page[r]);
    }
    printf("d) register + DMA + unregister:        %.3f ms (register alone %.3f)\n", (now() - t) / reps * 1e3, treg / reps * 1e3);
    t = now();
    for (int r = 0; r < reps; ++r) {
        hipMemcpyAsync(dev, pin, bytes, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
    }
    printf("e) DMA from pinned:                    %.3f ms (%.1f GB/s)\n", (now() - t) / reps * 1e3, bytes / ((now() - t) / reps) / 1e9);
    // D2H of a small result (1000 segments) into pinned vs pageable
    char* small = (char*)std::malloc(28000);
    t = now();
    for (int r = 0; r < 100; ++r) hipMemcpy(small, dev, 28000, hipMemcpyDeviceToHost);
    printf("f) D2H 28 KB to pageable (sync):       %.3f ms\n", (now() - t) / 100 * 1e3);
    t = now();
    for (int r = 0; r < 100; ++r) {
        hipMemcpyAsync(pin, dev, 28000, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
    }
    printf("g) D2H 28 KB to pinned (async + sync): %.3f ms\n", (now() - t) / 100 * 1e3);
    t = now();
    for (int r = 0; r < 100; ++r) hipStreamSynchronize(s);
    printf("h) empty stream synchronize:           %.4f ms\n", (now() - t) / 100 * 1e3);
    return 0;
}
