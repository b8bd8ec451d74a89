// Micro-benchmark for a design that was costed, not built: the GRU recurrence at large batch on bf16x3 operands (three bf16
// pieces per f32 value, six cross terms per product: DESIGN.md 3.0b) with the recurrent WEIGHTS stationary and the hidden state
// streamed -- workgroup (group, J) keeps unit tile J's three-piece R in registers for the whole launch (3 gates x 13 K-steps x
// 3 pieces = 117 fragments of 4 VGPRs, spread over its 8 wavefronts), the 25 workgroups of a group sit on one XCD
// (blockIdx % 8 under round-robin dispatch: speed only) and all read the same row tiles of h_{t-1} -- 16 sequences x 416 units
// as 13 K-steps x 3 pieces x 1 KB = 39 KB of MFMA fragments -- which one of them pulls into the XCD's L2 and the other 24 find
// there.  Per (row tile, unit tile): 39 KB into LDS, 3 gates x 13 K-steps x 6 cross terms = 234 v_mfma_f32_16x16x32_bf16,
// a cross-wavefront sum, the gate math, a 1.5 KB three-piece store.  This program times exactly that loop (random operand
// bits, no inter-workgroup flags: the workgroups of a group start together and drift) and prints the cycles per (row tile,
// unit tile), from which a launch of 49152 sequences x 53 steps follows: 3072 row tiles over G groups.
//   hipcc --offload-arch=gfx950 -O3 b3_ws_rate.hip -o b3_ws_rate && ./b3_ws_rate [row tiles per group] [steps]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
constexpr int KS = 13, NBLK = KS * 3; // K-steps of 32, blocks of 1 KB per row tile (3 pieces per K-step)

// MODE 0: everything; 1: no MFMAs; 2: no global loads (LDS contents reused); 3: loads only (no LDS reads, no MFMAs)
template <int MODE>
__global__ __launch_bounds__(512) void b3ws(const uint4* __restrict__ h, uint4* __restrict__ hout, const uint4* __restrict__ wfrag, int rt_per_group,
                                            int steps, float* sink)
{
    __shared__ __attribute__((aligned(16))) uint4 hb[2][NBLK * 64];   // 2 x 39 KB
    __shared__ __attribute__((aligned(16))) f32x4 part[8][3][64];     // per-wavefront partial sums of the three gates
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // groups 0..7: the 25 workgroups b = group + 8 J share an XCD under round-robin dispatch; groups 8, 9 (the CUs the eight
    // XCD-local groups leave over: 7 per XCD) are spread over all XCDs
    const int b = blockIdx.x;
    const int group = b < 200 ? b % 8 : 8 + (b - 200) / 25, J = b < 200 ? b / 8 : (b - 200) % 25;
    // this wavefront's K-steps: wave, wave + 8 (13 K-steps over 8 wavefronts), all 3 gates, 3 weight pieces each: 9 or 18 fragments
    bf16x8 w[2][3][3];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g = 0; g < 3; ++g)
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const uint4 v = wfrag[(((J * 2 + s) * 3 + g) * 3 + p) * 64 + lane];
                w[s][g][p] = __builtin_bit_cast(bf16x8, v);
            }
    const bool two = wave + 8 < KS;
    const uint4* hg = h + (size_t)group * rt_per_group * NBLK * 64;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};
    auto load = [&](int rt, int buf) { // 39 blocks over 8 wavefronts: 5 (4 for the last ones) x 1 KB each
        if (MODE == 2) return;
        for (int b = wave; b < NBLK; b += 8) hb[buf][b * 64 + lane] = hg[((size_t)rt * NBLK + b) * 64 + lane];
    };
    for (int st = 0; st < steps; ++st) {
        load(0, 0);
        __syncthreads();
        for (int rt = 0; rt < rt_per_group; ++rt) {
            const int buf = rt & 1;
            if (rt + 1 < rt_per_group) load(rt + 1, buf ^ 1);
            f32x4 acc[3] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (MODE != 3) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    if (s == 0 || two) {
                        const int ks = wave + 8 * s;
                        const bf16x8 xh = __builtin_bit_cast(bf16x8, hb[buf][(ks * 3 + 0) * 64 + lane]);
                        const bf16x8 xm = __builtin_bit_cast(bf16x8, hb[buf][(ks * 3 + 1) * 64 + lane]);
                        const bf16x8 xl = __builtin_bit_cast(bf16x8, hb[buf][(ks * 3 + 2) * 64 + lane]);
                        if (MODE != 1) {
#pragma unroll
                            for (int g = 0; g < 3; ++g) { // hh, hm, hl, mh, mm, lh
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][g][0], xh, acc[g], 0, 0, 0);
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][g][0], xm, acc[g], 0, 0, 0);
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][g][0], xl, acc[g], 0, 0, 0);
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][g][1], xh, acc[g], 0, 0, 0);
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][g][1], xm, acc[g], 0, 0, 0);
                                acc[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[s][g][2], xh, acc[g], 0, 0, 0);
                            }
                        } else {
                            acc[0][0] += __builtin_bit_cast(float, (int)xh[0] + (int)xm[1] + (int)xl[2]);
                        }
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < 3; ++g) part[wave][g][lane] = acc[g];
            __syncthreads(); // partial sums complete; row tile rt + 1 is in the other buffer; this buffer is free
            if (wave == (rt & 7)) { // the epilogue rotates over the wavefronts
                f32x4 z = {0.f, 0.f, 0.f, 0.f}, r = z, n = z;
#pragma unroll
                for (int v = 0; v < 8; ++v) { z += part[v][0][lane]; r += part[v][1][lane]; n += part[v][2][lane]; }
                f32x4 hnew;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zz = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-z[e]));
                    const float rr = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-r[e]));
                    const float nn = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(n[e] * rr));
                    hnew[e] = (1.0f - zz) * nn + zz * keep[e];
                }
                keep = hnew;
                // three-piece split and store: 16 x 16 values x 3 pieces x 2 B = 1.5 KB per (row tile, unit tile)
                uint32_t pk[6];
#pragma unroll
                for (int e = 0; e < 4; e += 2) {
                    float a = hnew[e], b = hnew[e + 1];
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const uint32_t ua = __builtin_bit_cast(uint32_t, a) & 0xFFFF0000u, ub = __builtin_bit_cast(uint32_t, b) & 0xFFFF0000u;
                        pk[p * 2 + e / 2] = (ua >> 16) | ub;
                        a -= __builtin_bit_cast(float, ua); b -= __builtin_bit_cast(float, ub);
                    }
                }
                uint2* o = reinterpret_cast<uint2*>(hout) + (((size_t)group * rt_per_group + rt) * 25 + J) * 3 * 64;
#pragma unroll
                for (int p = 0; p < 3; ++p) o[p * 64 + lane] = make_uint2(pk[p * 2], pk[p * 2 + 1]);
            }
        }
        __syncthreads();
    }
    if (keep[0] == 123.456f) sink[0] = keep[1];
}

int main(int argc, char** argv)
{
    const int rtg = argc > 1 ? atoi(argv[1]) : 384, steps = argc > 2 ? atoi(argv[2]) : 8;
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz = prop.clockRate * 1e-6;
    for (int groups : {8, 10}) {
        const size_t h_bytes = (size_t)groups * rtg * NBLK * 1024, o_bytes = (size_t)groups * rtg * 25 * 3 * 64 * 8;
        uint4 *h, *ho, *wf; float* sink;
        hipMalloc(&h, h_bytes); hipMalloc(&ho, o_bytes); hipMalloc(&wf, (size_t)25 * 2 * 9 * 1024); hipMalloc(&sink, 16);
        // operands: bf16 values in (-1, 1) with random mantissas (the bit pattern matters for the clock the chip holds)
        uint16_t* tmp = (uint16_t*)malloc(h_bytes);
        for (size_t i = 0; i < h_bytes / 2; ++i) tmp[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15));
        hipMemcpy(h, tmp, h_bytes, hipMemcpyHostToDevice);
        hipMemcpy(wf, tmp, (size_t)25 * 2 * 9 * 1024, hipMemcpyHostToDevice);
        free(tmp);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const char* names[4] = {"full loop", "no MFMAs", "no global loads", "loads only"};
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                const dim3 grid(groups * 25), block(512);
                if (mode == 0) hipLaunchKernelGGL(b3ws<0>, grid, block, 0, 0, h, ho, wf, rtg, steps, sink);
                if (mode == 1) hipLaunchKernelGGL(b3ws<1>, grid, block, 0, 0, h, ho, wf, rtg, steps, sink);
                if (mode == 2) hipLaunchKernelGGL(b3ws<2>, grid, block, 0, 0, h, ho, wf, rtg, steps, sink);
                if (mode == 3) hipLaunchKernelGGL(b3ws<3>, grid, block, 0, 0, h, ho, wf, rtg, steps, sink);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            const double us_per = best * 1e3 / ((double)steps * rtg);
            // a launch of 49152 sequences = 3072 row tiles over `groups` groups, 53 steps
            const double launch_ms = us_per * (3072.0 / groups) * 53 * 1e-3;
            printf("%2d groups x 25 workgroups, %d row tiles per group, %d steps: %-16s %7.3f us per (row tile, unit tile) = %5.0f clocks at %.2f GHz"
                   " -> %6.2f ms per 49152-sequence launch (the f32 kernel: 18.3 ms)\n", groups, rtg, steps, names[mode], us_per, us_per * 1e3 * ghz, ghz, launch_ms);
        }
        if (hipGetLastError() != hipSuccess) printf("HIP error\n");
        hipFree(h); hipFree(ho); hipFree(wf); hipFree(sink);
    }
    return 0;
}
