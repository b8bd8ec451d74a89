// Micro-benchmark: sustained v_mfma_f32_16x16x4_f32 issue rate on gfx950, nothing else in the loop.
// Gives the practical fp32-MFMA ceiling (clock/power management included) that the GRU and GEMM
// kernels' roofline fractions can be compared with.  Build: hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int NACC>
__global__ __launch_bounds__(512) void spin(float* out, int iters, float a, float b)
{
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, (float)i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < NACC; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    float* out;
    hipMalloc(&out, 4096 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int waves = 4; waves <= 8; waves += 4)
        for (int rep = 0; rep < 3; ++rep) {
            const int grid = 256;
            hipEventRecord(e0);
            hipLaunchKernelGGL((spin<16>), dim3(grid), dim3(64 * waves), 0, 0, out, iters, 1.0f, 0.5f);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            const double flop = (double)grid * waves * iters * 16 * 2048.0;
            printf("waves/CU=%d iters=%d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", waves, iters, ms, flop / ms * 1e-9,
                   flop / ms * 1e-9 / 157.3 * 100);
        }
    return 0;
}
