// Where do the workgroups of a second kernel go while a one-workgroup-per-CU kernel holds most of the chip?
//
// The pipelined recurrence of small launches (gru_ws2k_kernel: 228 workgroups of 1024 threads, 158 KB of LDS, all 512
// registers of every SIMD lane) leaves 28 of the 256 CUs idle for ~0.3 ms.  The streamed tail (nn_dispatch.cpp) puts GEMM
// workgroups of a second stream on them -- and measured 7x less throughput there than the MFMA count allows.  This tool
// takes the two kernels' SHAPES without their arithmetic:
//   hog     grid H x 1024 threads, 158 KB dynamic LDS, 128 VGPRs (launch bounds), spins for `hog_us` on s_memrealtime; with
//           traffic != 0 every wavefront also streams sc1 loads from a 64 MB buffer the way the recurrence's operand fetches do;
//   worker  grid W x 256 threads, 20 KB LDS: `mfma` dependent-free v_mfma_f32_16x16x4_f32 per wavefront (what a panel_gemm_s
//           workgroup issues: 400 for fc2, 608 for fc3) and `loads` 16-byte loads per lane per 16 MFMAs from a 4 MB buffer; it
//           records per workgroup: start / end (100 MHz ticks), XCC, SE, CU.
// Output: the worker kernel's duration alone and beside the hog, how many distinct CUs its workgroups ran on, the mean
// workgroups resident at once, and the per-workgroup duration -- i.e. whether the idle CUs are found at all (dispatch), and
// what a workgroup achieves there (interference).   ./idle_cu_census [H=228] [W=160] [mfma=400] [loads=1] [traffic=0] [hog_us=400]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <set>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Rec { unsigned long long t0, t1; unsigned hw; };

__device__ unsigned where_am_i()
{
    unsigned hw = 0, xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    return (hw & 0xFFFFFu) | ((xcc & 0xFu) << 24); // CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC in [27:24]
}

__global__ __launch_bounds__(1024) void hog(unsigned long long ticks, const f32x4* buf, size_t n4, int traffic, float* sink, Rec* hrec)
{
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (hrec && threadIdx.x == 0) { hrec[blockIdx.x].t0 = t0; hrec[blockIdx.x].hw = where_am_i(); }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    size_t i = ((size_t)blockIdx.x * 1024 + threadIdx.x) % n4;
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
        if (traffic) {
            // ~25 KB per workgroup every few microseconds, L1-bypassing like the recurrence's sc1 operand fetches
            if ((threadIdx.x >> 6) == 14) {
                for (int k = 0; k < 25; ++k) {
                    acc += __builtin_nontemporal_load(buf + i);
                    i = (i + 4099) % n4;
                }
            }
            __builtin_amdgcn_s_sleep(64);
        } else {
            __builtin_amdgcn_s_sleep(32);
        }
    }
    lds[threadIdx.x] = acc.x + acc.y;
    if (lds[(threadIdx.x + 1) & 1023] == 12345.f) *sink = acc.z;
}

__global__ __launch_bounds__(256) void worker(int mfma, int loads, const f32x4* buf, size_t n4, Rec* rec, float* sink)
{
    __shared__ float slab[5 * 1024];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0, a3 = a0, ld = a0;
    const float x = (float)threadIdx.x * 1e-3f, y = 1.0f + (float)blockIdx.x * 1e-4f;
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) % n4;
    for (int k = 0; k < mfma; k += 16) {
        for (int l = 0; l < loads; ++l) { ld += buf[i]; i = (i + 65537) % n4; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
        }
        slab[threadIdx.x] = ld.x;
        __syncthreads();
    }
    const f32x4 s = a0 + a1 + a2 + a3 + ld;
    if (s.x + s.y + s.z + s.w + slab[(threadIdx.x + 7) & 255] == 12345.678f) *sink = s.x;
    if (threadIdx.x == 0) {
        rec[blockIdx.x].t0 = t0;
        rec[blockIdx.x].t1 = __builtin_amdgcn_s_memrealtime();
        rec[blockIdx.x].hw = where_am_i();
    }
}

int main(int argc, char** argv)
{
    const int H = argc > 1 ? atoi(argv[1]) : 228, W = argc > 2 ? atoi(argv[2]) : 160, mfma = argc > 3 ? atoi(argv[3]) : 400;
    const int loads = argc > 4 ? atoi(argv[4]) : 1, traffic = argc > 5 ? atoi(argv[5]) : 0, hog_us = argc > 6 ? atoi(argv[6]) : 400;
    const size_t hbytes = 64u << 20, wbytes = 4u << 20;
    f32x4 *hbuf, *wbuf;
    float* sink;
    Rec *rec, *hrec;
    if (hipMalloc(&hrec, sizeof(Rec) * 1024) != hipSuccess) return 1;
    if (hipMalloc(&hbuf, hbytes) != hipSuccess || hipMalloc(&wbuf, wbytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess ||
        hipMalloc(&rec, sizeof(Rec) * W) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(hbuf, 0, hbytes);
    hipMemset(wbuf, 0, wbytes);
    hipStream_t s1, s2;
    hipStreamCreateWithFlags(&s1, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s2, hipStreamNonBlocking);
    const size_t lds = 158 * 1024;
    if (hipFuncSetAttribute((const void*)hog, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) { printf("attr failed\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    std::vector<Rec> h(W);
    auto report = [&](const char* what, float ms) {
        hipMemcpy(h.data(), rec, sizeof(Rec) * W, hipMemcpyDeviceToHost);
        unsigned long long lo = ~0ull, hi = 0, sum = 0;
        std::set<unsigned> cus;
        for (const Rec& r : h) { lo = std::min(lo, r.t0); hi = std::max(hi, r.t1); sum += r.t1 - r.t0; cus.insert(r.hw & 0x0F00FF00u); }
        std::vector<unsigned long long> d;
        for (const Rec& r : h) d.push_back(r.t1 - r.t0);
        std::sort(d.begin(), d.end());
        printf("%-34s kernel %8.1f us (first start -> last end %8.1f us); %3zu distinct CUs; %5.1f workgroups resident on average; "
               "per workgroup median %6.1f us, max %6.1f us\n", what, ms * 1e3, (hi - lo) * 0.01, cus.size(), (double)sum / (double)(hi - lo),
               d[d.size() / 2] * 0.01, d.back() * 0.01);
    };
    for (int rep = 0; rep < 2; ++rep) {
        // the worker alone
        hipEventRecord(e0, s2);
        hipLaunchKernelGGL(worker, dim3(W), dim3(256), 0, s2, mfma, loads, wbuf, wbytes / 16, rec, sink);
        hipEventRecord(e1, s2);
        hipStreamSynchronize(s2);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) report("worker alone", ms);
        // beside the hog: the hog first, the worker 50 us later (a host sleep would do; a tiny spin kernel keeps it on the device)
        hipLaunchKernelGGL(hog, dim3(H), dim3(1024), lds, s1, (unsigned long long)hog_us * 100ull, hbuf, hbytes / 16, traffic, sink, hrec);
        hipLaunchKernelGGL(hog, dim3(1), dim3(1024), lds, s2, 5000ull, hbuf, hbytes / 16, 0, sink, (Rec*)nullptr); // 50 us on the other stream
        hipEventRecord(e0, s2);
        hipLaunchKernelGGL(worker, dim3(W), dim3(256), 0, s2, mfma, loads, wbuf, wbytes / 16, rec, sink);
        hipEventRecord(e1, s2);
        hipStreamSynchronize(s2);
        hipStreamSynchronize(s1);
        hipEventElapsedTime(&ms, e0, e1);
        if (rep) report(traffic ? "worker beside the hog (+traffic)" : "worker beside the hog", ms);
        if (rep && getenv("CENSUS")) {
            // the map: per XCC and SE, which CUs the hog holds and where (and when) the worker's workgroups ran
            std::vector<Rec> hh(H);
            hipMemcpy(hh.data(), hrec, sizeof(Rec) * H, hipMemcpyDeviceToHost);
            unsigned long long w0 = ~0ull;
            for (const Rec& r : h) w0 = std::min(w0, r.t0);
            for (unsigned x = 0; x < 8; ++x)
                for (unsigned se = 0; se < 8; ++se) {
                    int nh = 0, nw = 0;
                    unsigned long long first = ~0ull, last = 0;
                    unsigned hog_cus = 0, wk_cus = 0;
                    for (const Rec& r : hh) if (((r.hw >> 24) & 15u) == x && ((r.hw >> 13) & 7u) == se) { ++nh; hog_cus |= 1u << ((r.hw >> 8) & 15u); }
                    for (const Rec& r : h) if (((r.hw >> 24) & 15u) == x && ((r.hw >> 13) & 7u) == se) { ++nw; wk_cus |= 1u << ((r.hw >> 8) & 15u); first = std::min(first, r.t0 - w0); last = std::max(last, r.t1 - w0); }
                    if (nh || nw) printf("  XCC %u SE %u: hog %2d workgroups (CU mask %04x), worker %3d workgroups (CU mask %04x) between %7.1f and %7.1f us\n", x, se, nh, hog_cus, nw, wk_cus, nw ? first * 0.01 : 0.0, nw ? last * 0.01 : 0.0);
                }
        }
    }
    printf("H = %d hog workgroups, W = %d worker workgroups, %d MFMAs per wavefront, %d loads per 16 MFMAs\n", H, W, mfma, loads);
    return hipGetLastError() != hipSuccess;
}
