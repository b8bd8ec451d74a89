// Micro-benchmark: per-CU global store drain rate on gfx950 for the shapes a 16x16 MFMA tile epilogue
// can produce.  Each wave issues float4 stores only; rows are 4800 B apart like the 1200-float GEMM output.
//   shape 0: 16 rows x 64 B per instruction (lane (m,q) -> row m, bytes 16q..)      <- the MFMA layout
//   shape 1:  4 rows x 256 B
//   shape 2:  1 row  x 1 KB (fully contiguous)
//   shape 3:  8 rows x 128 B, consecutive lanes on different rows (what a row_shr:8 DPP exchange gives)
//   shape 4:  8 rows x 128 B, 8 consecutive lanes per row
//   shape 5: 16 rows x 64 B, 4 consecutive lanes per row (shape 0's bytes after a 16x4 lane transpose)
// Build: hipcc --offload-arch=gfx950 -O3 store_rate.hip -o store_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int SHAPE>
__global__ __launch_bounds__(512) void store_kernel(float* C, int iters, size_t wg_stride)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    size_t off;
    if (SHAPE == 0) off = (size_t)(lane & 15) * 1200 + 4 * (lane >> 4);
    else if (SHAPE == 1) off = (size_t)(lane >> 4) * 1200 + 4 * (lane & 15);
    else if (SHAPE == 5) off = (size_t)(lane >> 2) * 1200 + 4 * (lane & 3);
    else if (SHAPE == 4) off = (size_t)(lane >> 3) * 1200 + 4 * (lane & 7);
    else if (SHAPE == 3) off = (size_t)(lane & 7) * 1200 + 16 * ((lane >> 3) & 1) + 4 * (lane >> 4);
    else off = 4 * lane;
    float* base = C + (size_t)blockIdx.x * wg_stride + (size_t)wave * 16 * 1200 + off;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    for (int it = 0; it < iters; ++it) {
        float* p = base + (size_t)it * (128 * 1200);
#pragma unroll
        for (int t = 0; t < 15; ++t) {
            if (SHAPE == 0 || SHAPE == 5) *reinterpret_cast<f32x4*>(p + 16 * t) = v;
            else if (SHAPE == 1) *reinterpret_cast<f32x4*>(p + 64 * (t % 4) + (size_t)(t / 4) * 4 * 1200) = v;
            else if (SHAPE == 3 || SHAPE == 4) *reinterpret_cast<f32x4*>(p + 32 * (t / 2) + (size_t)(t % 2) * 8 * 1200) = v;
            else *reinterpret_cast<f32x4*>(p + 256 * (t % 4) + (size_t)(t / 4) * 1200) = v;
        }
    }
}
template <int SHAPE> void run(float* C, int iters, size_t wg_stride, int waves_active, int grid = 256)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((store_kernel<SHAPE>), dim3(grid), dim3(64 * waves_active), 0, 0, C, iters, wg_stride);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double bytes = (double)grid * waves_active * iters * 15 * 1024.0;
        if (rep == 2)
            printf("shape %d, %d workgroups x %d waves: %.3f ms, %.2f TB/s chip, %.1f B/clk per active CU at 2.4 GHz\n", SHAPE, grid,
                   waves_active, ms, bytes / ms * 1e-9, bytes / ms * 1e-6 / grid / 2.4);
    }
}
int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 200;
    const size_t wg_stride = (size_t)iters * 128 * 1200 + 128 * 1200;
    float* C;
    if (hipMalloc(&C, 256 * wg_stride * 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    for (int w : {1, 2, 8}) {
        run<0>(C, iters, wg_stride, w);
        run<1>(C, iters, wg_stride, w);
        run<2>(C, iters, wg_stride, w);
    }
    // few active CUs: HBM is not the limit, this is the per-CU store path
    for (int g : {8, 32, 64}) {
        run<0>(C, iters, wg_stride, 8, g);
        run<1>(C, iters, wg_stride, 8, g);
        run<3>(C, iters, wg_stride, 8, g);
        run<4>(C, iters, wg_stride, 8, g);
        run<5>(C, iters, wg_stride, 8, g);
        run<2>(C, iters, wg_stride, 8, g);
    }
    return 0;
}
