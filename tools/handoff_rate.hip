// Micro-benchmark: what one step's all-to-all hand-off of the pipelined recurrence costs, without the arithmetic around it.
// Geometry of gru_ws2k_kernel's layer-1 loop: G groups of P workgroups (one per CU, 1024 threads), every workgroup is the
// producer of one 16 x 16 tile of h (1 KB) and the consumer of all P tiles of its group (P KB), 200 dependent steps:
//   step t: [wait for the peers' tiles of step t - 1] -> [a fixed stretch of "work"] -> publish the tile of step t.
// Two transports:
//   FLAG    (what the kernels do): sc1 tile store, s_waitcnt vmcnt(0), sc1 flag store; one wavefront polls the P flags
//           (timed first poll), then fetches the P tiles by sc1 LDS-DMA, waits, workgroup barrier.
//   GRANULE (the guide's R2 form at this size): the tile as 8-byte {value, tag} granules (2 KB, two 16-B sc1 stores per lane,
//           no drain, no flag); 15 wavefronts sweep the P x 2 KB with 16-B sc1 loads to registers after a timed wait, check
//           the tags (tag = step + epoch), retry what is stale, write the values to LDS, workgroup barrier.
//   LOCAL   (round 5): FLAG's protocol inside ONE XCD -- a group is the P workgroups that landed on the same XCC (each reads
//           HW_REG_XCC_ID and takes a ticket of its XCC for its tile index, so the grouping is a fact, not an assumption about
//           dispatch order), tile and flag are PLAIN stores (they stay in the XCD's L2; sc1 stores would drop the line), polls
//           and fetches are the same sc1 (L1-bypassing, L2-served) loads.  Needs P <= 32 and 8 P workgroups.
// Prints us per step for each, for several waits before the first poll / sweep.
//   hipcc --offload-arch=gfx950 -O3 handoff_rate.hip -o handoff_rate && ./handoff_rate [P] [G] [work_ticks]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
constexpr int AUX_SC1 = 16;

template <int MODE> // 0 FLAG, 1 GRANULE, 2 LOCAL
__global__ __launch_bounds__(1024) void handoff(float* hx, unsigned* flags, unsigned* err, int P, int steps, unsigned wait_ticks, unsigned work_ticks,
                                                unsigned epoch, float* sink, unsigned* xcc_tickets)
{
    constexpr bool GRANULE = MODE == 1, LOCAL = MODE == 2;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __attribute__((address_space(3))) char* lds3 = (__attribute__((address_space(3))) char*)smem;
    f32x4* hb = reinterpret_cast<f32x4*>(smem);             // P blocks of 64 float4
    volatile int& s_dead = *reinterpret_cast<volatile int*>(smem + 94 * 256); // (dynamic LDS only: a static array would shift the dynamic base's alignment)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int g = blockIdx.x / P, J = blockIdx.x - g * P;
    if (LOCAL) { // the group is the XCC this workgroup runs on, the tile index a ticket of that XCC
        __shared__ int s_j;
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        g = (int)(xcc & 7u);
        if (threadIdx.x == 0) s_j = (int)atomicAdd(xcc_tickets + g, 1u);
        __syncthreads();
        J = s_j;
        if (J >= P) return; // (more than P workgroups on this XCC: the launch's peers on another one will time out and say so)
    }
    const unsigned my_wg = (unsigned)(g * P + J);
    const unsigned tile_bytes = GRANULE ? 2048u : 1024u;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, (int)(2u * gridDim.x * tile_bytes), 0x00020000);
    __attribute__((address_space(1))) unsigned* my_flag = (__attribute__((address_space(1))) unsigned*)(flags + g * 64 + J);
    __attribute__((address_space(1))) unsigned* poll_flag = (__attribute__((address_space(1))) unsigned*)(flags + g * 64 + (lane < P ? lane : 0));
    if (tid == 0) s_dead = 0;
    __syncthreads();
    f32x4 acc = {0.f, 0.f, 0.f, (float)J};
    for (int t = 0; t < steps; ++t) {
        // ---- publish the tile of step t (wavefront 12, like the helper)
        if (wave == 12) {
            const f32x4 h = acc + (f32x4){(float)t, 1.f, 2.f, 3.f};
            const unsigned slot = (unsigned)((t & 1) * gridDim.x + my_wg) * tile_bytes;
            if (GRANULE) {
                const unsigned tag = epoch + (unsigned)t + 1u;
                const u32x4 a = {__builtin_bit_cast(unsigned, h[0]), tag, __builtin_bit_cast(unsigned, h[1]), tag};
                const u32x4 b = {__builtin_bit_cast(unsigned, h[2]), tag, __builtin_bit_cast(unsigned, h[3]), tag};
                __builtin_amdgcn_raw_buffer_store_b128(a, rs, (unsigned)lane * 32u, slot, AUX_SC1);
                __builtin_amdgcn_raw_buffer_store_b128(b, rs, (unsigned)lane * 32u + 16u, slot, AUX_SC1);
            } else if (LOCAL) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs, (unsigned)lane * 16u, slot, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); // a plain store
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs, (unsigned)lane * 16u, slot, AUX_SC1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        // ---- acquire the peers' tiles of step t
        const unsigned long long t_start = __builtin_amdgcn_s_memrealtime();
        const unsigned row0 = (unsigned)((t & 1) * gridDim.x + g * P) * tile_bytes;
        if (GRANULE) {
            if (wave < 15) {
                while (__builtin_amdgcn_s_memrealtime() - t_start < wait_ticks) __builtin_amdgcn_s_sleep(1);
                const unsigned tag = epoch + (unsigned)t + 1u;
                for (int S = wave; S < P; S += 15) {
                    for (;;) {
                        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 32u, row0 + S * 2048u, AUX_SC1);
                        const u32x4 b = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)lane * 32u + 16u, row0 + S * 2048u, AUX_SC1);
                        const bool ok = a[1] == tag && a[3] == tag && b[1] == tag && b[3] == tag;
                        if (__all(ok)) {
                            hb[S * 64 + lane] = (f32x4){__builtin_bit_cast(float, a[0]), __builtin_bit_cast(float, a[2]), __builtin_bit_cast(float, b[0]), __builtin_bit_cast(float, b[2])};
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        if (__builtin_amdgcn_s_memrealtime() - t_start > 2000000ull) { if (lane == 0) { s_dead = 1; *err = 1u; } break; }
                    }
                }
            }
        } else if (wave == 14) {
            while (__builtin_amdgcn_s_memrealtime() - t_start < wait_ticks) __builtin_amdgcn_s_sleep(1);
            for (;;) {
                const unsigned v = lane < P ? __hip_atomic_load(poll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
                if (__all(v >= (unsigned)(t + 1))) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t_start > 2000000ull) { if (lane == 0) { s_dead = 1; *err = 1u; } break; }
            }
            for (int S = 0; S < P; ++S)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds3 + S * 1024), 16, (unsigned)lane * 16u, row0 + S * 1024u, 0, AUX_SC1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        if (s_dead) return;
        // ---- the step's "work": every wavefront reads the operands (as the chains do) and the gate wavefronts idle for work_ticks
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int S = wave & 1; S < P; S += 2) s += hb[S * 64 + lane];
        acc += s * 1e-6f;
        const unsigned long long w0 = __builtin_amdgcn_s_memrealtime();
        while (__builtin_amdgcn_s_memrealtime() - w0 < work_ticks) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
    }
    if (acc[0] == 1234.5f) sink[0] = acc[1];
}

int main(int argc, char** argv)
{
    const int P = argc > 1 ? atoi(argv[1]) : 13, G = argc > 2 ? atoi(argv[2]) : 6, work = argc > 3 ? atoi(argv[3]) : 300;
    const int steps = 200, n_wg = P * G;
    float *hx, *sink; unsigned *flags, *err, *tickets;
    hipMalloc(&hx, (size_t)2 * n_wg * 2048); hipMalloc(&flags, 64 * 64 * 4); hipMalloc(&err, 4); hipMalloc(&sink, 16); hipMalloc(&tickets, 64);
    hipMemset(hx, 0, (size_t)2 * n_wg * 2048);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const size_t lds = 96 * 1024; // one workgroup per CU
    hipFuncSetAttribute((const void*)handoff<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)handoff<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipFuncSetAttribute((const void*)handoff<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    unsigned epoch = 1000;
    for (int granule = 0; granule < 3; ++granule) {
        if (granule == 2 && (G != 8 || P > 32)) { printf("LOCAL needs G = 8 (one group per XCC) and P <= 32\n"); continue; }
        for (unsigned wait : {0u, 20u, 40u, 80u, 120u, 160u, 200u, 240u, 280u}) {
            float best = 1e30f; unsigned bad = 0;
            for (int rep = 0; rep < 4; ++rep) {
                hipMemset(flags, 0, 64 * 64 * 4); hipMemset(err, 0, 4); hipMemset(tickets, 0, 64);
                epoch += 1000;
                hipEventRecord(e0);
                if (granule == 1) hipLaunchKernelGGL(handoff<1>, dim3(n_wg), dim3(1024), lds, 0, hx, flags, err, P, steps, wait, (unsigned)work, epoch, sink, tickets);
                else if (granule == 2) hipLaunchKernelGGL(handoff<2>, dim3(n_wg), dim3(1024), lds, 0, hx, flags, err, P, steps, wait, (unsigned)work, epoch, sink, tickets);
                else hipLaunchKernelGGL(handoff<0>, dim3(n_wg), dim3(1024), lds, 0, hx, flags, err, P, steps, wait, (unsigned)work, epoch, sink, tickets);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); bad |= e;
                if (ms < best) best = ms;
            }
            printf("P=%2d G=%d work %4.1f us %-8s first poll after %4.1f us: %6.2f us per step (hand-off %5.2f us)%s\n", P, G, work * 0.01, granule == 2 ? "LOCAL" : granule ? "GRANULE" : "FLAG",
                   wait * 0.01, best * 1e3 / steps, best * 1e3 / steps - work * 0.01, bad ? "  TIMEOUT" : "");
        }
    }
    return 0;
}
