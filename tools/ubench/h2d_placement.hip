// Where should the staging of pageable frames run?  A pipeline of 64 frames of 33 MB (3840x2160 fp32) from pageable memory
// through a ring of four page-locked staging buffers into HBM -- T threads copy frame i+1 while the DMA engine sends frame i
// -- under a placement matrix: source frames first touched on the GPU's NUMA node or on another, staging buffers allocated
// from a thread on either, copy threads bound to either.  Prints the sustained rate of each combination, next to the
// floor (DMA from page-locked memory alone).  (profiles/r04_h2d_paths.txt)
// Build: hipcc -O2 --offload-arch=gfx950 -o h2d_placement h2d_placement.hip -lpthread
#include <hip/hip_runtime.h>
#include <immintrin.h>
#include <pthread.h>
#include <sched.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static std::vector<int> cpu_list(const std::string& text) {
    std::vector<int> out;
    size_t i = 0;
    while (i < text.size()) {
        size_t j = text.find(',', i);
        if (j == std::string::npos) j = text.size();
        const std::string part = text.substr(i, j - i);
        const size_t d = part.find('-');
        if (!part.empty() && part[0] >= '0' && part[0] <= '9') {
            const int a = std::atoi(part.c_str()), b = d == std::string::npos ? a : std::atoi(part.c_str() + d + 1);
            for (int c = a; c <= b; ++c) out.push_back(c);
        }
        i = j + 1;
    }
    return out;
}
static std::vector<int> node_cpus(int node) {
    std::ifstream f("/sys/devices/system/node/node" + std::to_string(node) + "/cpulist");
    std::string s;
    std::getline(f, s);
    std::vector<int> all = cpu_list(s), ok;
    cpu_set_t cur;
    sched_getaffinity(0, sizeof(cur), &cur);  // (cores this process may use at all)
    for (int c : all)
        if (CPU_ISSET(c, &cur)) ok.push_back(c);
    return ok;
}
static void bind_to(const std::vector<int>& cpus, int k = -1) {
    if (cpus.empty()) return;
    cpu_set_t set;
    CPU_ZERO(&set);
    if (k >= 0) CPU_SET(cpus[(size_t)k % cpus.size()], &set);
    else
        for (int c : cpus) CPU_SET(c, &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
}

static void nt_copy(char* d, const char* s, size_t n) {
    size_t blocks = n / 64;
    while (blocks--) {
        const __m128i a = _mm_loadu_si128((const __m128i*)s), b = _mm_loadu_si128((const __m128i*)(s + 16));
        const __m128i c = _mm_loadu_si128((const __m128i*)(s + 32)), e = _mm_loadu_si128((const __m128i*)(s + 48));
        _mm_stream_si128((__m128i*)d, a);
        _mm_stream_si128((__m128i*)(d + 16), b);
        _mm_stream_si128((__m128i*)(d + 32), c);
        _mm_stream_si128((__m128i*)(d + 48), e);
        s += 64;
        d += 64;
    }
    _mm_sfence();
}

struct Crew {
    std::vector<std::thread> th;
    std::atomic<uint32_t> job{0};
    std::atomic<int> left{0};
    std::atomic<bool> quit{false};
    char* dst = nullptr;
    const char* src = nullptr;
    size_t bytes = 0;
    int T = 1;
    bool nt = true;
    void piece(int t) {
        const size_t per = (bytes / T + 4095) & ~(size_t)4095, b = (size_t)t * per, e = std::min(bytes, b + per);
        if (b < e) {
            if (nt) nt_copy(dst + b, src + b, e - b);
            else std::memcpy(dst + b, src + b, e - b);
        }
        left.fetch_sub(1);
    }
    void start(int T_, const std::vector<int>& cpus) {
        T = T_;
        for (int t = 1; t < T; ++t)
            th.emplace_back([this, t, cpus] {
                bind_to(cpus, t);
                uint32_t last = 0;
                while (!quit.load()) {
                    const uint32_t g = job.load(std::memory_order_acquire);
                    if (g == last) {
                        std::this_thread::yield();
                        continue;
                    }
                    last = g;
                    piece(t);
                }
            });
    }
    void run(char* d, const char* s, size_t n) {
        dst = d;
        src = s;
        bytes = n;
        left.store(T);
        job.fetch_add(1, std::memory_order_release);
        piece(0);
        while (left.load() > 0) std::this_thread::yield();
    }
    ~Crew() {
        quit.store(true);
        for (auto& t : th) t.join();
    }
};

int main(int argc, char** argv) {
    const size_t bytes = (size_t)3840 * 2160 * 4;
    const int n_src = 16, n_frames = 64, R = 4;
    char bus[64] = {0};
    hipDeviceGetPCIBusId(bus, sizeof(bus), 0);
    int gpu_node = -1, n_nodes = 0;
    {
        std::string b(bus);
        for (auto& ch : b) ch = (char)tolower(ch);
        std::ifstream f("/sys/bus/pci/devices/" + b + "/numa_node");
        if (f) f >> gpu_node;
        while (std::ifstream("/sys/devices/system/node/node" + std::to_string(n_nodes) + "/cpulist").good()) ++n_nodes;
    }
    if (gpu_node < 0) gpu_node = 0;
    const int other = n_nodes > 1 ? (gpu_node + n_nodes / 2) % n_nodes : gpu_node;
    const std::vector<int> cpus_gpu = node_cpus(gpu_node), cpus_other = node_cpus(other);
    printf("GPU %s on NUMA node %d of %d (%zu usable cores there); 'other' = node %d (%zu usable cores)\n", bus, gpu_node, n_nodes, cpus_gpu.size(), other,
           cpus_other.size());
    char* dev;
    hipMalloc((void**)&dev, bytes * R);
    hipStream_t s;
    hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    hipEvent_t ev[R];
    for (auto& e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    auto alloc_on = [&](const std::vector<int>& cpus, bool pinned, unsigned flags) {
        char* p = nullptr;
        std::thread([&] {
            bind_to(cpus);
            if (pinned) {
                if (hipHostMalloc((void**)&p, bytes, flags) != hipSuccess) p = nullptr;
            } else {
                p = (char*)std::malloc(bytes);
            }
            if (p) std::memset(p, 1, bytes);  // first touch here
        }).join();
        return p;
    };
    // floor: DMA from page-locked memory alone
    {
        char* pin = alloc_on(cpus_gpu, true, hipHostMallocDefault);
        hipMemcpy(dev, pin, bytes, hipMemcpyHostToDevice);
        const double t0 = now();
        for (int i = 0; i < n_frames; ++i) hipMemcpyAsync(dev + (size_t)(i % R) * bytes, pin, bytes, hipMemcpyHostToDevice, s);
        hipStreamSynchronize(s);
        printf("DMA from page-locked memory alone: %.1f GB/s\n", n_frames * bytes / (now() - t0) / 1e9);
        hipHostFree(pin);
    }
    const int Ts[] = {4, 8, 16};
    for (int src_other = 0; src_other < (n_nodes > 1 ? 2 : 1); ++src_other)
        for (int stage_other = 0; stage_other < (n_nodes > 1 ? 2 : 1); ++stage_other)
            for (int thr_other = 0; thr_other < (n_nodes > 1 ? 2 : 1); ++thr_other) {
                const auto& c_src = src_other ? cpus_other : cpus_gpu;
                const auto& c_stage = stage_other ? cpus_other : cpus_gpu;
                const auto& c_thr = thr_other ? cpus_other : cpus_gpu;
                if (c_src.empty() || c_stage.empty() || c_thr.empty()) continue;
                std::vector<char*> srcs(n_src), stage(R);
                for (auto& p : srcs) p = alloc_on(c_src, false, 0);
                for (auto& p : stage) p = alloc_on(c_stage, true, hipHostMallocNumaUser);
                for (int T : Ts)
                    for (int nt = 1; nt >= 0; --nt) {
                        double best = 0;
                        std::thread([&] {
                            bind_to(c_thr, 0);
                            Crew crew;
                            crew.nt = nt != 0;
                            crew.start(T, c_thr);
                            for (int rep = 0; rep < 3; ++rep) {
                                const double t0 = now();
                                for (int i = 0; i < n_frames; ++i) {
                                    const int slot = i % R;
                                    if (i >= R) hipEventSynchronize(ev[slot]);
                                    crew.run(stage[slot], srcs[i % n_src], bytes);
                                    hipMemcpyAsync(dev + (size_t)slot * bytes, stage[slot], bytes, hipMemcpyHostToDevice, s);
                                    hipEventRecord(ev[slot], s);
                                }
                                hipStreamSynchronize(s);
                                best = std::max(best, n_frames * bytes / (now() - t0) / 1e9);
                            }
                        }).join();
                        printf("frames on %-5s staging on %-5s threads on %-5s  T=%2d %-8s %.1f GB/s\n", src_other ? "other" : "gpu", stage_other ? "other" : "gpu",
                               thr_other ? "other" : "gpu", T, nt ? "nt-store" : "memcpy", best);
                        fflush(stdout);
                    }
                for (auto p : srcs) std::free(p);
                for (auto p : stage) hipHostFree(p);
            }
    return 0;
}
