// Micro-benchmark: sustained v_mfma_f32_16x16x32_f16 rate on gfx950 for the instruction mixes of kernels_h3.hip.
//   mode 0: 16 independent accumulators, operands fixed in registers
//   mode 1: the GEMM's tile pattern: per tile 6 MFMAs on 2 accumulators (dependent at distance 2), A operands change per tile
//   mode 2: mode 1 + the tile's two ds_read_b128 (ring of 3 tiles, counted lgkmcnt waits)
// Operands are random f16 values (zeros read high: clocks).  Build: hipcc --offload-arch=gfx950 -O3 mfma_f16_rate.hip -o mfma_f16_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

template <int MODE>
__global__ __launch_bounds__(512) void spin(const f32x4* in, float* out, int iters)
{
    __shared__ __attribute__((aligned(16))) float lds[15 * 512];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 15 * 128; i += blockDim.x) reinterpret_cast<f32x4*>(lds)[i] = in[i % 1024];
    __syncthreads();
    h16x8 xh[2], xl[2];
    for (int r = 0; r < 2; ++r) {
        xh[r] = __builtin_bit_cast(h16x8, in[lane + 64 * r]);
        xl[r] = __builtin_bit_cast(h16x8, in[lane + 128 + 64 * r]);
    }
    f32x4 acc[2][15];
    for (int t = 0; t < 15; ++t) acc[0][t] = acc[1][t] = (f32x4){0.f, 0.f, 0.f, (float)t};
    h16x8 w[6];
    for (int i = 0; i < 6; ++i) w[i] = __builtin_bit_cast(h16x8, in[256 + lane + 64 * i]);
    typedef __attribute__((address_space(3))) float lds_float;
    const unsigned rd = (unsigned)(uintptr_t)(lds_float*)lds + lane * 16u;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int t = 0; t < 15; ++t) {
                    acc[0][t] = MFMA_H(w[0], xh[0], acc[0][t]);
                    acc[1][t] = MFMA_H(w[1], xh[1], acc[1][t]);
                }
        } else {
            f32x4 rh[3], rl[3];
            if (MODE == 2) {
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rh[k]) : "v"(rd), "n"(0));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rl[k]) : "v"(rd), "n"(1024));
                }
            }
#pragma unroll
            for (int t = 0; t < 15; ++t) {
                h16x8 ah, al;
                if (MODE == 2) {
                    asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(rh[t % 3]), "+v"(rl[t % 3]));
                    ah = __builtin_bit_cast(h16x8, rh[t % 3]);
                    al = __builtin_bit_cast(h16x8, rl[t % 3]);
                } else {
                    ah = w[(2 * t) % 6];
                    al = w[(2 * t + 1) % 6];
                }
                acc[0][t] = MFMA_H(al, xh[0], acc[0][t]);
                acc[1][t] = MFMA_H(al, xh[1], acc[1][t]);
                acc[0][t] = MFMA_H(ah, xl[0], acc[0][t]);
                acc[1][t] = MFMA_H(ah, xl[1], acc[1][t]);
                acc[0][t] = MFMA_H(ah, xh[0], acc[0][t]);
                acc[1][t] = MFMA_H(ah, xh[1], acc[1][t]);
                if (MODE == 2) {
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rh[t % 3]) : "v"(rd), "n"(2048));
                    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(rl[t % 3]) : "v"(rd), "n"(3072));
                }
            }
        }
    }
    f32x4 s = acc[0][0];
    for (int t = 0; t < 15; ++t) s += acc[0][t] + acc[1][t];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y + s.z + s.w;
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 3000;
    f32x4* in;
    float* out;
    hipMalloc(&in, 1024 * 16);
    hipMalloc(&out, 4096 * 512 * 4);
    unsigned short h[1024 * 8];
    srand(1);
    for (int i = 0; i < 1024 * 8; ++i) h[i] = (unsigned short)(0x3000 + (rand() & 0x0FFF) + ((rand() & 1) << 15)); // ~0.1 .. 1, random sign
    hipMemcpy(in, h, sizeof h, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode)
        for (int waves = 4; waves <= 8; waves += 4)
            for (int rep = 0; rep < 2; ++rep) {
                const int grid = 256;
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL((spin<0>), dim3(grid), dim3(64 * waves), 0, 0, in, out, iters);
                if (mode == 1) hipLaunchKernelGGL((spin<1>), dim3(grid), dim3(64 * waves), 0, 0, in, out, iters);
                if (mode == 2) hipLaunchKernelGGL((spin<2>), dim3(grid), dim3(64 * waves), 0, 0, in, out, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                const double n_mfma = (double)grid * waves * iters * 90;
                const double flop = n_mfma * 16384.0;
                printf("mode %d waves/CU=%d: %.3f ms  %.0f TFLOP/s f16 (%.1f%% of 2500); ns per MFMA per SIMD %.2f\n", mode, waves, ms,
                       flop / ms * 1e-9, flop / ms * 1e-9 / 2500 * 100, ms * 1e6 / (n_mfma / (grid * 4)));
            }
    return 0;
}
