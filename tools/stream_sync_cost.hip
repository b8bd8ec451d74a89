// What does it cost to order two streams against each other around a long kernel?  (the streamed tail's fork and join)
//   A  plain: k1 -> long -> k2 on one stream (reference gaps)
//   B  event fork: k1, hipEventRecord, long on s1; s2 waits for the event, runs w; hipEventRecord on s2; s1 waits, k2
//   C  hipStreamWaitValue32: s2's w waits (in the command processor) for a word the long kernel writes when it is resident;
//      s1's k2 waits for a word w writes at its end
// Kernels stamp s_memrealtime at start and end; printed: gaps k1 end -> long start, long end -> k2 start, long start -> w start.
#include <hip/hip_runtime.h>
#include <stdio.h>
struct Stamp { unsigned long long t0, t1; };
__global__ void small(Stamp* s, unsigned* flag, unsigned val)
{
    if (threadIdx.x == 0) { s->t0 = __builtin_amdgcn_s_memrealtime(); }
    __builtin_amdgcn_s_sleep(100);
    if (threadIdx.x == 0) {
        if (flag) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        s->t1 = __builtin_amdgcn_s_memrealtime();
    }
}
__global__ __launch_bounds__(1024) void longk(Stamp* s, unsigned* flag, unsigned val, unsigned long long ticks)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) { s->t0 = t0; if (flag) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
    if (threadIdx.x == 0 && blockIdx.x == 0) s->t1 = __builtin_amdgcn_s_memrealtime();
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main()
{
    Stamp* st; unsigned* flags;
    CK(hipMalloc(&st, sizeof(Stamp) * 8));
    CK(hipMalloc(&flags, 64));
    hipStream_t s1, s2;
    CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    hipEvent_t ef, ej;
    CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming));
    CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
    int can = 0;
    hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0);
    printf("hipDeviceAttributeCanUseStreamWaitValue = %d\n", can);
    Stamp h[8];
    for (int mode = 0; mode < 3; ++mode) {
        if (mode == 2 && !can) break;
        double g1 = 0, g2 = 0, g3 = 0;
        const int reps = 20;
        for (int r = 0; r < reps + 2; ++r) {
            CK(hipMemsetAsync(flags, 0, 64, s1));
            CK(hipStreamSynchronize(s1));
            hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s1, st + 0, (unsigned*)nullptr, 0u);
            if (mode == 1) { CK(hipEventRecord(ef, s1)); CK(hipStreamWaitEvent(s2, ef, 0)); }
            hipLaunchKernelGGL(longk, dim3(200), dim3(1024), 0, s1, st + 1, flags, 1u, 20000ull); // 200 us
            if (mode == 2) CK(hipStreamWaitValue32(s2, flags, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
            if (mode >= 1) {
                hipLaunchKernelGGL(small, dim3(28), dim3(256), 0, s2, st + 2, flags + 1, 1u);
                if (mode == 1) { CK(hipEventRecord(ej, s2)); CK(hipStreamWaitEvent(s1, ej, 0)); }
                else CK(hipStreamWaitValue32(s1, flags + 1, 1u, hipStreamWaitValueGte, 0xFFFFFFFFu));
            }
            hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s1, st + 3, (unsigned*)nullptr, 0u);
            CK(hipStreamSynchronize(s1));
            CK(hipStreamSynchronize(s2));
            CK(hipMemcpy(h, st, sizeof(h), hipMemcpyDeviceToHost));
            if (r >= 2) {
                g1 += (double)(h[1].t0 - h[0].t1) * 0.01; g2 += (double)(h[3].t0 - h[1].t1) * 0.01;
                if (mode >= 1) g3 += (double)((long long)h[2].t0 - (long long)h[1].t0) * 0.01;
            }
        }
        printf("%s: k1 end -> long start %6.2f us   long end -> k2 start %6.2f us   long start -> side kernel start %7.2f us\n",
               mode == 0 ? "A one stream      " : mode == 1 ? "B event fork/join " : "C stream wait value", g1 / reps, g2 / reps, g3 / reps);
    }
    return 0;
}
