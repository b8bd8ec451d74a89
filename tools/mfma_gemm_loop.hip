// Micro-benchmark: the inner loop of panel_gemm3_kernel with its pieces switched on one by one, to see
// which of them keeps the MFMA pipe below its pure-issue ceiling (tools/mfma_peak.hip: 99 %).
//   mode 0: 15 tiles x 8 MFMAs per step, operands in registers
//   mode 1: + ds_read_b128 of the next step's tile after each tile, lgkmcnt(14) waits
//   mode 2: + workgroup barrier every 5 steps
//   mode 3: + LDS-DMA of a 75 KB slab every 5 steps (double buffered)
//   mode 4: + one global float4 activation load per step per row tile (rows L2-resident)
//   mode 14: like 4, but the first activation load of a phase is issued before that phase's slab DMA
//   mode 5: + accumulator stores (245 KB per workgroup) every 5 phases to fresh HBM rows, accumulators reset
//   mode 7/8: like 5 but each store instruction covers 8 rows x 128 B / 4 rows x 256 B instead of 16 rows x 64 B
//             (timing only: the values land in the wrong places)
//   mode 9..13: like 5 but no activation load in the first 1..5 steps after the stores (vmcnt is in-order:
//             a load issued after the stores cannot be waited for without waiting for the stores)
//   mode 6: + activation rows change every 5 phases (HBM-cold, like the real row panels)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
template <int T, int N> struct StaticFor {
    template <class F> static __device__ __forceinline__ void run(F&& f) { f(std::integral_constant<int, T>{}); StaticFor<T + 1, N>::run(f); }
};
template <int N> struct StaticFor<N, N> { template <class F> static __device__ __forceinline__ void run(F&&) {} };
template <int OFF> __device__ __forceinline__ void lds_read_b128(f32x4& dst, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF)); }
template <int N> __device__ __forceinline__ void lds_wait(f32x4& dst) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(N)); }

constexpr int NT = 15, SP = 5;
template <int MODE>
__global__ __launch_bounds__(512) void loop_kernel(const float* __restrict__ W, const float* __restrict__ A, float* out, int phases, float* __restrict__ C)
{
    __shared__ __attribute__((aligned(16))) float slab[2][NT * SP * 256];
    typedef __attribute__((address_space(3))) float lds_float;
    typedef const __attribute__((address_space(1))) f32x4* gptr4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 2 * NT * SP * 256; i += 512) slab[0][i] = W[i % (NT * SP * 256)];
    __syncthreads();
    const unsigned slab_addr[2] = {(unsigned)(uintptr_t)(lds_float*)slab[0] + lane * 16u, (unsigned)(uintptr_t)(lds_float*)slab[1] + lane * 16u};
    f32x4 acc[2][NT];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const float* a_ptr[2] = {A + ((size_t)(blockIdx.x * 16 + wave * 2) * 16 + (lane & 15)) * 400 + 4 * (lane >> 4),
                             A + ((size_t)(blockIdx.x * 16 + wave * 2 + 1) * 16 + (lane & 15)) * 400 + 4 * (lane >> 4)};
    f32x4 a0[2] = {*(gptr4)a_ptr[0], *(gptr4)a_ptr[1]};
    int buf = 0;
    for (int p = 0; p < phases; ++p) {
        f32x4 a1pre[2];
        if (MODE == 14) {
            // the first step's activation load goes out before the slab DMA, so that waiting for it at the end
            // of that step (vmcnt retires in order) does not also wait for the DMA
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) a1pre[rt] = *(gptr4)(a_ptr[rt] + 16 * ((p * SP + 1) % 25));
        }
        if (MODE >= 3) {
            const float* src = W;
            float* dst = slab[buf ^ 1];
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                int b = wave + i * 8;
                b = b < NT * SP ? b : NT * SP - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b * 256 + lane * 4),
                                                 (__attribute__((address_space(3))) void*)(dst + b * 256), 16, 0, 0);
            }
        }
        unsigned rd = slab_addr[MODE >= 3 ? buf : 0];
        f32x4 w[NT];
        if (MODE >= 1) StaticFor<0, NT>::run([&](auto tc) { constexpr int t = decltype(tc)::value; lds_read_b128<t * 1024>(w[t], rd); });
        else {
#pragma unroll
            for (int t = 0; t < NT; ++t) w[t] = (f32x4){1.f + t, 0.5f, 0.25f, 2.f};
        }
        auto do_step = [&](int s, auto pre) {
            f32x4 a1[2];
            if (decltype(pre)::value) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) a1[rt] = a1pre[rt];
            } else if (MODE >= 4 && !(MODE >= 9 && MODE < 14 && (p % 5) == 0 && s < MODE - 8)) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) a1[rt] = *(gptr4)(a_ptr[rt] + 16 * ((p * SP + s + 1) % 25));
            } else {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) a1[rt] = a0[rt];
            }
            const unsigned rdn = (s + 1 < SP) ? rd : rd - NT * 1024;
            StaticFor<0, NT>::run([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if (MODE >= 1) lds_wait<NT - 1>(w[t]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][t] = MFMA16(w[t].x, a0[rt].x, acc[rt][t]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][t] = MFMA16(w[t].y, a0[rt].y, acc[rt][t]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][t] = MFMA16(w[t].z, a0[rt].z, acc[rt][t]);
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) acc[rt][t] = MFMA16(w[t].w, a0[rt].w, acc[rt][t]);
                if (MODE >= 1) lds_read_b128<(NT + t) * 1024>(w[t], rdn);
            });
            rd += NT * 1024;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) a0[rt] = a1[rt];
        };
        if (MODE == 14) {
            do_step(0, std::true_type{});
            for (int s = 1; s < SP; ++s) do_step(s, std::false_type{});
        } else {
            for (int s = 0; s < SP; ++s) do_step(s, std::false_type{});
        }
        if (MODE >= 2) __syncthreads();
        if (MODE >= 3) buf ^= 1;
        if (MODE >= 5 && MODE != 14 && p % 5 == 4) {
            const size_t item = (size_t)(p / 5) * 256 + blockIdx.x;
#pragma unroll
            for (int rt = 0; rt < 2; ++rt) {
                if (MODE == 7) {
                    // pairs of tiles: instruction 2k covers rows 0-7, 2k+1 rows 8-15, 32 columns each
                    float* c = C + ((item * 16 + wave * 2 + rt) * 16 + (lane & 7)) * 240 + 4 * (lane >> 4) + 16 * ((lane >> 3) & 1);
#pragma unroll
                    for (int t = 0; t < NT; ++t) { *reinterpret_cast<f32x4*>(c + 32 * (t >> 1) + (t & 1) * 8 * 240) = acc[rt][t]; acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                } else if (MODE == 8) {
                    float* c = C + ((item * 16 + wave * 2 + rt) * 16 + (lane & 3)) * 240 + 4 * (lane >> 4) + 16 * ((lane >> 2) & 3);
#pragma unroll
                    for (int t = 0; t < NT; ++t) { *reinterpret_cast<f32x4*>(c + 64 * (t >> 2) + (t & 3) * 4 * 240) = acc[rt][t]; acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                } else {
                float* c = C + ((item * 16 + wave * 2 + rt) * 16 + (lane & 15)) * 240 + 4 * (lane >> 4);
#pragma unroll
                for (int t = 0; t < NT; ++t) { *reinterpret_cast<f32x4*>(c + 16 * t) = acc[rt][t]; acc[rt][t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                }
            }
            if (MODE == 6) {
#pragma unroll
                for (int rt = 0; rt < 2; ++rt) a_ptr[rt] += (size_t)256 * 256 * 400;
            }
        }
    }
    f32x4 sum = acc[0][0];
#pragma unroll
    for (int t = 1; t < NT; ++t) sum += acc[0][t] + acc[1][t];
    out[blockIdx.x * 512 + tid] = sum.x + sum.y + sum.z + sum.w;
}
template <int MODE> void run(const float* W, const float* A, float* out, int phases, float* C = nullptr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((loop_kernel<MODE>), dim3(256), dim3(512), 0, 0, W, A, out, phases, C);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 256.0 * 8 * phases * SP * NT * 8 * 2048.0;
        if (rep) printf("mode %d: %.3f ms  %.1f TFLOP/s (%.1f%% of 157.3)\n", MODE, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3 * 100);
    }
}
int main(int argc, char** argv)
{
    const int phases = argc > 1 ? atoi(argv[1]) : 1000;
    float *W, *A, *out;
    hipMalloc(&W, NT * SP * 256 * 4 * 2);
    hipMalloc(&A, (size_t)256 * 256 * 400 * 4);
    hipMalloc(&out, 256 * 512 * 4);
    // operand data matter: an all-zero run draws less power and holds a higher clock
    const bool zero = argc > 2 && atoi(argv[2]) == 0;
    {
        const size_t nw = NT * SP * 256 * 2, na = (size_t)256 * 256 * 400;
        float* h = (float*)malloc(na * 4);
        unsigned x = 12345u;
        for (size_t i = 0; i < na; ++i) { x = x * 1664525u + 1013904223u; h[i] = zero ? 0.f : ((int)(x >> 8) - (1 << 23)) * (1.0f / (1 << 23)) * 0.05f; }
        hipMemcpy(W, h, nw * 4, hipMemcpyHostToDevice);
        hipMemcpy(A, h, na * 4, hipMemcpyHostToDevice);
        free(h);
    }
    run<0>(W, A, out, phases);
    run<1>(W, A, out, phases);
    run<2>(W, A, out, phases);
    run<3>(W, A, out, phases);
    run<4>(W, A, out, phases);
    run<14>(W, A, out, phases);
    {
        const size_t items = (size_t)(phases / 5 + 1);
        float *C, *A2;
        if (hipMalloc(&C, items * 256 * 256 * 240 * 4) != hipSuccess) return 1;
        run<5>(W, A, out, phases, C);
        run<9>(W, A, out, phases, C);
        run<10>(W, A, out, phases, C);
        run<11>(W, A, out, phases, C);
        run<13>(W, A, out, phases, C);
        if (hipMalloc(&A2, items * 256 * 256 * 400 * 4) != hipSuccess) return 1;
        hipMemset(A2, 0, items * 256 * 256 * 400 * 4);
        run<6>(W, A2, out, phases, C);
    }
    return 0;
}
