// drain_rate.hip -- where the device -> host drain of fvad_engine_run goes: the D2H DMA into page-locked slots, the host memcpy out of
// them into pageable memory by N threads (and the reverse direction's two halves), each alone: the ceilings behind
// extra.pcie_inclusive_* (docs/LAB_NOTES.md, round 5).   ./drain_rate [MB per block = 6] [blocks = 32]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void par_copy(char* dst, const char* src, size_t block, int n_blocks, int n_threads)
{
    auto work = [&](int t) { for (int i = t; i < n_blocks; i += n_threads) memcpy(dst + (size_t)i * block, src + (size_t)i * block, block); };
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
}

int main(int argc, char** argv)
{
    const size_t block = (size_t)(argc > 1 ? atoi(argv[1]) : 6) << 20;
    const int n_blocks = argc > 2 ? atoi(argv[2]) : 32;
    const size_t total = block * n_blocks;
    char *dev, *pin, *page;
    CK(hipMalloc((void**)&dev, total));
    CK(hipMemset(dev, 1, total));
    CK(hipHostMalloc((void**)&pin, total, hipHostMallocDefault));
    page = (char*)malloc(total);
    memset(page, 2, total);
    memset(pin, 3, total);
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (int rep = 0; rep < 2; ++rep) {
        double t0 = now();
        for (int i = 0; i < n_blocks; ++i) CK(hipMemcpyAsync(pin + i * block, dev + i * block, block, hipMemcpyDeviceToHost, st));
        CK(hipStreamSynchronize(st));
        double dt = now() - t0;
        if (rep) printf("D2H DMA into page-locked memory, %d x %zu MB:   %6.1f GB/s\n", n_blocks, block >> 20, total / dt / 1e9);
        t0 = now();
        for (int i = 0; i < n_blocks; ++i) CK(hipMemcpyAsync(dev + i * block, pin + i * block, block, hipMemcpyHostToDevice, st));
        CK(hipStreamSynchronize(st));
        dt = now() - t0;
        if (rep) printf("H2D DMA from page-locked memory:                 %6.1f GB/s\n", total / dt / 1e9);
    }
    for (int nt : {1, 2, 4, 8, 16}) {
        double best_out = 0, best_in = 0;
        for (int rep = 0; rep < 3; ++rep) {
            double t0 = now(); par_copy(page, pin, block, n_blocks, nt); double dt = now() - t0;
            best_out = std::max(best_out, total / dt / 1e9);
            t0 = now(); par_copy(pin, page, block, n_blocks, nt); dt = now() - t0;
            best_in = std::max(best_in, total / dt / 1e9);
        }
        printf("%2d threads: page-locked -> pageable %6.1f GB/s   pageable -> page-locked %6.1f GB/s\n", nt, best_out, best_in);
    }
    {   // what asking the runtime about a pointer costs (the engine asks per lane whether user memory is page-locked)
        hipPointerAttribute_t attr;
        double t0 = now();
        int n_host = 0;
        for (int i = 0; i < 256; ++i) { if (hipPointerGetAttributes(&attr, page + (size_t)i * 4096) == hipSuccess && attr.type == hipMemoryTypeHost) n_host++; else (void)hipGetLastError(); }
        double dt = now() - t0;
        printf("hipPointerGetAttributes on pageable memory: %.1f us per call\n", dt / 256 * 1e6);
        t0 = now();
        for (int i = 0; i < 256; ++i) { if (hipPointerGetAttributes(&attr, pin + (size_t)i * 4096) == hipSuccess && attr.type == hipMemoryTypeHost) n_host++; }
        dt = now() - t0;
        printf("hipPointerGetAttributes on page-locked memory: %.1f us per call (%d)\n", dt / 256 * 1e6, n_host);
    }
    // the drain as the engine does it: DMA of wave w beside the memcpy of wave w - 1 (16 blocks per wave, 8 threads)
    for (int nt : {8, 16}) {
        hipEvent_t ev[2]; CK(hipEventCreate(&ev[0])); CK(hipEventCreate(&ev[1]));
        const int per = 16, n_waves = (n_blocks + per - 1) / per;
        double t0 = now();
        for (int w = 0; w <= n_waves; ++w) {
            if (w < n_waves) {
                for (int i = w * per; i < std::min(n_blocks, (w + 1) * per); ++i) CK(hipMemcpyAsync(pin + i * block, dev + i * block, block, hipMemcpyDeviceToHost, st));
                CK(hipEventRecord(ev[w & 1], st));
            }
            if (w >= 1) {
                CK(hipEventSynchronize(ev[(w - 1) & 1]));
                const int b0 = (w - 1) * per, nb = std::min(per, n_blocks - b0);
                par_copy(page + b0 * block, pin + b0 * block, block, nb, nt);
            }
        }
        double dt = now() - t0;
        printf("pipelined drain, %2d threads:                     %6.1f GB/s\n", nt, total / dt / 1e9);
    }
    return 0;
}
