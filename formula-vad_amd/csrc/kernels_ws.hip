// kernels_ws.hip -- GRU recurrence for small batches, weight-stationary across the chip.
//
// A streaming push of a few 0.5 s chunks (and BASELINE config 3's literal 82-chunk batch) is a latency
// problem: 53 dependent steps per GRU layer with a [n_seq x 400] x [400 x 1200] product each.
// gru_lat_kernel gives 16 sequences to one workgroup, which then re-streams the whole 1.92 MB of
// recurrent weights from L2 every step (~38 us per step) while 250 CUs idle.  Here the WEIGHTS stay
// put and the hidden state travels:
//   * workgroup (g, J) owns unit tile J (16 of the 400 hidden units) for the row tiles (16 sequences
//     each) of sequence group g; its three gate wavefronts keep their 25 KB of R (one gate x 25
//     super-steps of tile J) in 100 VGPRs for the whole kernel -- no weight traffic after the prologue;
//   * per step the workgroup fetches h_{t-1} of a row tile once (25 KB, the MFMA B operands of all three
//     gates) into LDS, each gate wavefront runs 100 MFMAs on it and hands its 16x16 tile to the owner
//     wavefront through LDS; the owner does the gate math and publishes the tile of h_t: one 1 KB
//     write-through (sc1) store in exactly the layout every other workgroup reads as its operand for
//     super-step S = J;
//   * the 25 workgroups of a group exchange h_t all-to-all through HBM-side memory: per-workgroup
//     monotonic flags (value = steps published), polled by one wavefront (25 lanes, one flag each),
//     payload read with sc1 loads -- the write-through / sc1 hand-off of the CDNA4 guide (every storing
//     wave drains vmcnt, workgroup barrier, ONE lane stores the flag; the polling wave joins a barrier
//     before the other waves load).  h buffers alternate by step parity: a workgroup can only be one step
//     ahead of the slowest member of its group, because it needs that member's h_t to go on.
// All workgroups of a launch must be resident together (spin waits): the grid is at most 25 * (CUs / 25)
// workgroups of one per CU (dynamic LDS is padded above half a CU's), every spin is bounded by a wall-clock
// deadline, and a workgroup that gives up raises *err and exits; gru_lat_kernel is then run for the layer
// by the caller (it is launched behind this kernel with *err as its guard and returns at once otherwise).
//
// Semantics: zero h at row 0 of every sequence (src/NSNet2.zig:57-58,71-112), gate order z,r,h,
// linear_before_reset = 1; gi holds Wx + Wb, Rb is added here (the convention of gru_lat_kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "kernels.h"
#include "nn_device.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

// Diagnostics build (make -C formula-vad_amd/csrc diag -> libfvad_hip_diag.so, for tools/ws2_variants.py and
// tools/ws2_trace.py): only there do the TIMING-ONLY bits of the context option ws2_variant (1, 2, 4, 32: wrong results) and
// the step trace (64) exist.  The shipping library compiles them out and fvad_ctx_set_option rejects them.
#ifndef FVAD_DIAG
#define FVAD_DIAG 0
#endif
#define WS_DIAG(variant_, bits_) (FVAD_DIAG != 0 && ((variant_) & (bits_)))

constexpr int WS_AUX_SC1 = 16;                            // buffer cache-policy bit: sc1 (system-coherent level 1)
// wait before a step's first poll, in 10 ns ticks (measured by building with other values: 634 -> 571 us for 130 sequences,
// 1060 -> 1014 for 300, 1210 -> 1170 for 600): gru_ws_kernel; gru_ws2_kernel per row tile of a group on top of a base
constexpr unsigned WS_WAIT_TICKS = 80, WS2_WAIT_BASE = 40, WS2_WAIT_PER_RT = 40;

template <int OWN>
__global__ __launch_bounds__(256) void gru_ws_kernel(const float* __restrict__ gi, const float* __restrict__ R2frag,
                                                     const float* __restrict__ bR, float* __restrict__ hout,
                                                     float* hx, unsigned* flags, unsigned* err, int T, int RT, int n_rt,
                                                     int gi_js, int gi_gs, unsigned long long spin_ticks)
{
    // gi_js / gi_gs: floats between unit tiles / between gates in a gi row (16, 400 gate-major; 48, 16 tile-major)
    // dynamic LDS only (a static array would shift the 16-byte alignment of the dynamic base):
    // hbuf[2][25][64] float4 (h_{t-1} of a row tile, double-buffered), xch[RT][3 gates][64] float4, one int
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* hbuf = reinterpret_cast<f32x4*>(smem);
    f32x4* xch = hbuf + 2 * GRU_J * 64;
    volatile int* s_dead = reinterpret_cast<volatile int*>(xch + (size_t)RT * 3 * 64);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    const int g = blockIdx.x / GRU_J;
    const int J = blockIdx.x - g * GRU_J;

    // a workgroup of an earlier layer gave up: the whole network pass falls back to gru_lat_kernel
    if (tid == 0) *s_dead = (int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (*s_dead) return;
    __syncthreads();

    // ---- prologue: this wavefront's stationary weights (gate = wave, tile J, all 25 super-steps)
    f32x4 w[GRU_J];
    if (wave < 3) {
        const f32x4* src = reinterpret_cast<const f32x4*>(R2frag) + ((size_t)(J * 3 + wave) * GRU_J) * 64 + lane;
#pragma unroll
        for (int S = 0; S < GRU_J; ++S) w[S] = src[S * 64];
    } else {
#pragma unroll
        for (int S = 0; S < GRU_J; ++S) w[S] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    const f32x4 bz = *reinterpret_cast<const f32x4*>(bR + 16 * J + 4 * q);
    const f32x4 br = *reinterpret_cast<const f32x4*>(bR + GRU_H + 16 * J + 4 * q);
    const f32x4 bn = *reinterpret_cast<const f32x4*>(bR + 2 * GRU_H + 16 * J + 4 * q);

    const auto rs = __builtin_amdgcn_make_buffer_rsrc(hx, 0, 2 * n_rt * GRU_J * 1024, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    __attribute__((address_space(1))) unsigned* my_flag = (__attribute__((address_space(1))) unsigned*)(flags + g * GRU_J + J);
    __attribute__((address_space(1))) unsigned* poll_flag =
        (__attribute__((address_space(1))) unsigned*)(flags + g * GRU_J + (lane < GRU_J ? lane : 0));

    // row tiles whose gate math this wavefront owns: ((wave + 1) & 3) + 4 o  (wave 3, which has no MFMA work,
    // owns row tile 0)
    int own_rt[OWN];
    bool own_ok[OWN];
    size_t own_row[OWN];
    f32x4 hp[OWN];
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        own_rt[o] = ((wave + 1) & 3) + 4 * o;
        const int rtg = g * RT + own_rt[o];
        own_ok[o] = own_rt[o] < RT && rtg < n_rt;
        own_row[o] = (size_t)(own_ok[o] ? rtg : 0) * 16 + m;
        hp[o] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    auto publish = [&](int o, int t, f32x4 h) {
        const int rtg = g * RT + own_rt[o];
        const unsigned off = (unsigned)((((t & 1) * n_rt + rtg) * GRU_J + J) * 1024);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs, lane16, off, WS_AUX_SC1);
        *reinterpret_cast<f32x4*>(hout + (own_row[o] * T + t) * GRU_H + 16 * J + 4 * q) = h;
    };

    // ---- t = 0: h_{-1} = 0, so R h + Rb = Rb
#pragma unroll
    for (int o = 0; o < OWN; ++o) {
        if (!own_ok[o]) continue;
        const float* gp = gi + (own_row[o] * T) * (3 * GRU_H) + gi_js * J + 4 * q;
        const f32x4 giz = *reinterpret_cast<const f32x4*>(gp);
        const f32x4 gir = *reinterpret_cast<const f32x4*>(gp + gi_gs);
        const f32x4 gin = *reinterpret_cast<const f32x4*>(gp + 2 * gi_gs);
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = fast_sigmoid(giz[r] + bz[r]);
            const float rr = fast_sigmoid(gir[r] + br[r]);
            const float n = fast_tanh(gin[r] + rr * bn[r]);
            h[r] = (1.0f - z) * n + z * 0.0f;
        }
        publish(o, 0, h);
        hp[o] = h;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the flag is raised
    __syncthreads();
    if (tid == 0) __hip_atomic_store(my_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    for (int t = 1; t < T; ++t) {
        // gate operands of this step: independent of h, requested before the wait
        f32x4 giz[OWN], gir[OWN], gin[OWN];
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            if (own_ok[o]) {
                const float* gp = gi + (own_row[o] * T + t) * (3 * GRU_H) + gi_js * J + 4 * q;
                giz[o] = *reinterpret_cast<const f32x4*>(gp);
                gir[o] = *reinterpret_cast<const f32x4*>(gp + gi_gs);
                gin[o] = *reinterpret_cast<const f32x4*>(gp + 2 * gi_gs);
            }
        }
        // ---- wait until every workgroup of the group has published h_{t-1} (flag >= t).  The first poll is timed: the
        // peers' flags cannot be visible before ~WS_WAIT_TICKS x 10 ns after this workgroup raised its own, and a poll
        // before that is a wasted round trip and traffic on the group's flag line (gru_ws2k_kernel)
        if (wave == 3) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < WS_WAIT_TICKS) __builtin_amdgcn_s_sleep(1);
            int dead = 0;
            for (;;) {
                const unsigned v = __hip_atomic_load(poll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= (unsigned)t)) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) { dead = 1; break; }
            }
            if (lane == 0) {
                if (dead) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = dead;
            }
        }
        __syncthreads();
        if (*s_dead) return; // uniform: every wavefront reads the same word after the barrier

        // ---- recurrent product of this step.  h_{t-1} of a row tile (25 KB) is fetched ONCE per workgroup: the
        // four wavefronts load 6-7 of its 25 super-step blocks each (sc1: straight from the memory side) and
        // park them in LDS, where the three gate wavefronts read their B operands; the next row tile's loads
        // are in flight while this one's 300 MFMAs run (double-buffered, one barrier per row tile).
        {
            const int pprev = (t - 1) & 1;
            const int my_rt = (n_rt - g * RT < RT) ? n_rt - g * RT : RT;
            constexpr int PER = (GRU_J + 3) / 4;
            f32x4 ld[PER];
            auto issue = [&](int rt) {
                const unsigned base = (unsigned)(((pprev * n_rt + g * RT + rt) * GRU_J) * 1024);
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int S = wave + 4 * i;
                    if (S < GRU_J)
                        ld[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, base + S * 1024, WS_AUX_SC1));
                }
            };
            issue(0);
            for (int rt = 0; rt < my_rt; ++rt) {
                f32x4* hb = hbuf + (rt & 1) * (GRU_J * 64);
#pragma unroll
                for (int i = 0; i < PER; ++i) {
                    const int S = wave + 4 * i;
                    if (S < GRU_J) hb[S * 64 + lane] = ld[i];
                }
                __syncthreads(); // this row tile's h is in LDS; the other buffer's readers (row tile rt - 1) are done
                if (rt + 1 < my_rt) issue(rt + 1);
                if (wave < 3) {
                    // two accumulation chains (even / odd super-steps): a dependent MFMA issues every 40 cycles,
                    // an independent one every 32
                    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                    for (int S = 0; S < GRU_J; ++S) {
                        const f32x4 hv = hb[S * 64 + lane];
                        if (S & 1) {
                            a1 = MFMA16(w[S].x, hv.x, a1);
                            a1 = MFMA16(w[S].y, hv.y, a1);
                            a1 = MFMA16(w[S].z, hv.z, a1);
                            a1 = MFMA16(w[S].w, hv.w, a1);
                        } else {
                            a0 = MFMA16(w[S].x, hv.x, a0);
                            a0 = MFMA16(w[S].y, hv.y, a0);
                            a0 = MFMA16(w[S].z, hv.z, a0);
                            a0 = MFMA16(w[S].w, hv.w, a0);
                        }
                    }
                    xch[(rt * 3 + wave) * 64 + lane] = a0 + a1;
                }
            }
        }
        __syncthreads();

        // ---- gate math + publish by the owners
#pragma unroll
        for (int o = 0; o < OWN; ++o) {
            if (!own_ok[o]) continue;
            const f32x4 az = xch[(own_rt[o] * 3 + 0) * 64 + lane];
            const f32x4 ar = xch[(own_rt[o] * 3 + 1) * 64 + lane];
            const f32x4 an = xch[(own_rt[o] * 3 + 2) * 64 + lane];
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fast_sigmoid(giz[o][r] + (az[r] + bz[r]));
                const float rr = fast_sigmoid(gir[o][r] + (ar[r] + br[r]));
                const float n = fast_tanh(gin[o][r] + rr * (an[r] + bn[r]));
                h[r] = (1.0f - z) * n + z * hp[o][r];
            }
            publish(o, t, h);
            hp[o] = h;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ------------------------------------------------------------------ both GRU layers in one launch, pipelined
// gru_ws_kernel runs the two layers one after the other: 2 x 54 exchange-bound steps (~5.3 us each) with the second
// layer's input projection (a GEMM launch) between them -- 0.63 of BASELINE config 3's 0.85 ms at its literal
// 82-chunk batch.  Layer 2's step t needs only h1_t and h2_{t-1}, so it can run beside layer 1's step t + 1: here a
// group of 38 workgroups serves one set of row tiles, 13 for layer 1 and 25 for layer 2, and the chain is 55 steps.
//   * a workgroup has 8 wavefronts in two halves (0-2 / 4-6 gate wavefronts, 3 / 7 helpers); every gate wavefront
//     keeps one gate's 25 fragment blocks of one matrix tile in 100 VGPRs for the whole launch.  A layer-1 workgroup
//     owns TWO unit tiles of R1 (2 pair, 2 pair + 1, one per half; the 13th workgroup's second half idles).  A
//     layer-2 workgroup owns ONE unit tile and both of its matrices -- W_ih in the first half, R in the second: layer 2
//     computes its own input projection gi2_t = W_ih h1_t instead of reading it from a GEMM (both matrices of two
//     tiles would be 200 VGPRs per wavefront: spills);
//   * the hand-off is gru_ws_kernel's (sc1 write-through payload, every storing wave drains vmcnt, workgroup barrier,
//     one lane raises the workgroup's monotonic flag; one wavefront polls, the others load after a barrier it joins),
//     with one flag per workgroup (13 + 25 per group) and two exchange rings: h1 in FOUR slots (step % 4) because
//     layer 1 may run ahead of its reader -- it waits until every layer-2 workgroup of its group has published h2 of
//     step t - 4, i.e. has consumed h1_{t-4}, before it overwrites that slot -- and h2 in two (step parity);
//   * layer 2's step t: wait for h1_t and h2_{t-1} -> fetch both -> W_ih h1_t on the first half's gate wavefronts
//     BESIDE R h2_{t-1} on the second half's -> gates -> publish h2_t: the input projection costs a second 25 KB fetch,
//     not a second product phase (a first version did the two products one after the other: 7.6 us per step
//     against 5.3 us for gru_ws_kernel).
// Bounded spins and the error word as in gru_ws_kernel; behind this kernel the engine queues gru_ws2_fallback_kernel
// (kernels_nn.hip): gru_lat's body for layer 1, layer 2's input projection and gru_lat's body for layer 2, per 16
// sequences, run only if the error word was raised; it also counts the pass and zeroes the polled words.
// Arithmetic: per output the same two accumulation chains (even / odd super-steps) as gru_ws_kernel; gi2 is
// (W_ih h1 chains) + Wb instead of a GEMM's bias-last chain, so this family differs from the others in the last bits.
__global__ __launch_bounds__(512) void gru_ws2_kernel(const float* __restrict__ gi1, const float* __restrict__ R1frag,
                                                      const float* __restrict__ bR1, const float* __restrict__ W2frag,
                                                      const float* __restrict__ bW2, const float* __restrict__ R2frag,
                                                      const float* __restrict__ bR2, float* __restrict__ hout2,
                                                      float* hx1, float* hx2, unsigned* flags1, unsigned* flags2,
                                                      unsigned* err, int T, int RT, int n_rt, unsigned long long spin_ticks,
                                                      int variant)
{
    // variant != 0: TIMING-ONLY builds of the step (wrong results; tools/ws2_variants.py, context option ws2_variant):
    // 1 layer 2 without its input projection (no h1 fetch, no W_ih product); 2 no row-major store of h2;
    // 4 layer 1 alone (layer 2's workgroups exit, no back-pressure)
    // dynamic LDS, in float4s: hbuf[2][2][25][64] (one row tile of h1 and one of h2, double-buffered); per (row tile, tile of the pair):
    // xch[3 gates][64] recurrent products, xci[3][64] layer 2's input projection / layer 1's gi of this step,
    // hpv[64] the previous h of the tile; one int
    extern __shared__ __attribute__((aligned(16))) float smem[];
    f32x4* hbuf = reinterpret_cast<f32x4*>(smem);
    f32x4* xch = hbuf + 4 * GRU_J * 64;
    f32x4* xci = xch + (size_t)RT * 6 * 64;
    f32x4* hpv = xci + (size_t)RT * 6 * 64;
    volatile int* s_dead = reinterpret_cast<volatile int*>(hpv + (size_t)RT * 2 * 64);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ws = wave >> 2;  // half of the workgroup
    const int wg = wave & 3;   // gate wavefront (0..2) or the half's helper (3)
    const int q = lane >> 4;
    const int g = blockIdx.x / 38;
    const int r38 = blockIdx.x - g * 38;
    const int layer = r38 >= 13;
    const int pair = layer ? r38 - 13 : r38;   // layer 1: pair of unit tiles; layer 2: the unit tile
    const int J = layer ? pair : 2 * pair + ws;
    const bool tile_ok = J < GRU_J; // layer 1's 13th workgroup: its second half has no tile, it only loads and joins barriers
    const int Jc = tile_ok ? J : 0;
    const int tslot = layer ? 0 : ws; // where this wavefront's tile sits in xch / xci / hpv
    const int my_rt = (n_rt - g * RT < RT) ? n_rt - g * RT : RT;
    // who does the gate math, fetches gi and publishes: layer 1: each half's helper for its tile; layer 2: wavefront 7
    const bool helper = layer ? wave == 7 : (wg == 3 && tile_ok);

    if (tid == 0) *s_dead = (int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int i = tid; i < RT * 2 * 64; i += 512) hpv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < RT * 6 * 64; i += 512) xci[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    if (*s_dead) return;
    if (WS_DIAG(variant, 4) && layer) return;
    __syncthreads();

    // ---- stationary weights of a gate wavefront: gate wg of tile J, all 25 super-steps: R1 (layer 1), or layer 2's
    // W_ih (first half) / R2 (second half)
    f32x4 w[GRU_J];
    {
        const bool gate = wg < 3 && tile_ok;
        const size_t blk = ((size_t)(Jc * 3 + (wg < 3 ? wg : 0)) * GRU_J) * 64 + lane;
        const f32x4* src = reinterpret_cast<const f32x4*>(layer ? (ws ? R2frag : W2frag) : R1frag) + blk;
#pragma unroll
        for (int S = 0; S < GRU_J; ++S) w[S] = gate ? src[S * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    // biases of this wavefront's tile, held in registers (a load per step would sit on the step's critical path):
    // Rb, and for layer 2 the input projection's Wb
    const float* bR = (layer ? bR2 : bR1) + 16 * Jc + 4 * q;
    const float* bW = bW2 + 16 * Jc + 4 * q;
    const f32x4 bz = *reinterpret_cast<const f32x4*>(bR), br = *reinterpret_cast<const f32x4*>(bR + GRU_H),
                bn = *reinterpret_cast<const f32x4*>(bR + 2 * GRU_H);
    f32x4 wbz = (f32x4){0.f, 0.f, 0.f, 0.f}, wbr = wbz, wbn = wbz;
    if (layer) {
        wbz = *reinterpret_cast<const f32x4*>(bW);
        wbr = *reinterpret_cast<const f32x4*>(bW + GRU_H);
        wbn = *reinterpret_cast<const f32x4*>(bW + 2 * GRU_H);
    }
    // layer 1's gi of the step for the helper's first two row tiles: requested BEFORE the wait, like gru_ws_kernel
    // does (further row tiles are fetched inside the product phase, beside the MFMAs, through xci)
    constexpr int GPRE = 2;
    f32x4 gpre[GPRE][3];
    auto request_gi = [&](int t) {
        if (layer == 0 && helper) {
#pragma unroll
            for (int rt = 0; rt < GPRE; ++rt) {
                if (rt < my_rt) {
                    const size_t row = (size_t)(g * RT + rt) * 16 + (lane & 15);
                    const float* gp = gi1 + (row * T + t) * (3 * GRU_H) + 48 * J + 4 * q; // tile-major rows: [25 J][3 gates][16]
                    gpre[rt][0] = *reinterpret_cast<const f32x4*>(gp);
                    gpre[rt][1] = *reinterpret_cast<const f32x4*>(gp + 16);
                    gpre[rt][2] = *reinterpret_cast<const f32x4*>(gp + 32);
                }
            }
        }
    };

    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(hx1, 0, 4 * n_rt * GRU_J * 1024, 0x00020000);
    const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(hx2, 0, 2 * n_rt * GRU_J * 1024, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    __attribute__((address_space(1))) unsigned* my_flag =
        (__attribute__((address_space(1))) unsigned*)((layer ? flags2 + g * 25 : flags1 + g * 13) + pair);
    // the polling wavefront (wave 3): lanes 0..12 watch layer 1's 13 flags of this group, lanes 32..56 layer 2's 25
    __attribute__((address_space(1))) unsigned* poll_flag =
        (__attribute__((address_space(1))) unsigned*)(lane < 32 ? flags1 + g * 13 + (lane < 13 ? lane : 0)
                                                                : flags2 + g * 25 + (lane - 32 < 25 ? lane - 32 : 0));

    // wait until layer 1 has published `need1` steps and layer 2 `need2` (0 = no requirement); uniform false on a deadline
    auto wait_for = [&](unsigned need1, unsigned need2) -> bool {
        if (wave == 3) {
            const unsigned need = lane < 32 ? need1 : need2;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < WS2_WAIT_BASE + WS2_WAIT_PER_RT * (unsigned)RT) __builtin_amdgcn_s_sleep(1); // timed first poll (gru_ws_kernel)
            int dead = 0;
            for (;;) {
                const unsigned v = __hip_atomic_load(poll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= need)) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) { dead = 1; break; }
            }
            if (lane == 0) {
                if (dead) __hip_atomic_store((__attribute__((address_space(1))) unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = dead;
            }
        }
        __syncthreads();
        return *s_dead == 0; // uniform: every wavefront reads the same word after the barrier
    };

    // Product of the gate wavefronts' stationary fragments with every row tile of the group's h in ring `rs` at byte
    // offset slot_base (of row tile 0 of the batch): a row tile's h is fetched once per workgroup (25 KB: the eight
    // wavefronts load 3-4 of its 25 blocks each, sc1) into LDS, double-buffered, one barrier per row tile.  While the
    // gate wavefronts run their 100 MFMAs the helper wavefronts of layer 1 fetch the step's gi (gi_t >= 0) into xci.
    // Layer 1: both halves multiply their own tile of R1 with h1_{t-1} (source A).  Layer 2: the first half multiplies
    // W_ih with h1 (source A), the second half R2 with h2 (source B) -- at the same time when both are asked for.
    // useA / useB: which sources are fetched and which halves compute (layer 1: A only, both halves).
    constexpr int PER = (GRU_J + 7) / 8;
    f32x4 ldA[PER], ldB[PER];
    auto issue = [&](int rt, bool doA, unsigned slotA, bool doB, unsigned slotB) {
        const unsigned row0 = (unsigned)((g * RT + rt) * GRU_J) * 1024u;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int S = wave + 8 * i;
            if (S < GRU_J) {
                if (doA) ldA[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs1, lane16, slotA + row0 + S * 1024, WS_AUX_SC1));
                if (doB) ldB[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs2, lane16, slotB + row0 + S * 1024, WS_AUX_SC1));
            }
        }
    };
    auto product = [&](bool useA, unsigned slotA, bool useB, unsigned slotB, int gi_t) {
        issue(0, useA, slotA, useB, slotB);
        for (int rt = 0; rt < my_rt; ++rt) {
            f32x4* hbA = hbuf + (rt & 1) * (2 * GRU_J * 64);
            f32x4* hbB = hbA + GRU_J * 64;
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int S = wave + 8 * i;
                if (S < GRU_J) {
                    if (useA) hbA[S * 64 + lane] = ldA[i];
                    if (useB) hbB[S * 64 + lane] = ldB[i];
                }
            }
            __syncthreads();
            if (rt + 1 < my_rt) issue(rt + 1, useA, slotA, useB, slotB);
            if (wg < 3) {
                const bool mine = layer ? (ws ? useB : useA) : true; // layer 2: this half's matrix was asked for
                if (tile_ok && mine) {
                    const f32x4* hb = (layer && ws) ? hbB : hbA;
                    f32x4 a0 = (f32x4){0.f, 0.f, 0.f, 0.f}, a1 = a0;
#pragma unroll
                    for (int S = 0; S < GRU_J; ++S) {
                        const f32x4 hv = hb[S * 64 + lane];
                        if (S & 1) {
                            a1 = MFMA16(w[S].x, hv.x, a1);
                            a1 = MFMA16(w[S].y, hv.y, a1);
                            a1 = MFMA16(w[S].z, hv.z, a1);
                            a1 = MFMA16(w[S].w, hv.w, a1);
                        } else {
                            a0 = MFMA16(w[S].x, hv.x, a0);
                            a0 = MFMA16(w[S].y, hv.y, a0);
                            a0 = MFMA16(w[S].z, hv.z, a0);
                            a0 = MFMA16(w[S].w, hv.w, a0);
                        }
                    }
                    // layer 1: R1 h1 -> xch; layer 2: W_ih h1 -> xci (first half), R2 h2 -> xch (second half)
                    f32x4* xdst = (layer && !ws) ? xci : xch;
                    xdst[((rt * 2 + tslot) * 3 + wg) * 64 + lane] = a0 + a1;
                }
            } else if (gi_t >= 0 && helper && rt >= GPRE) {
                const size_t row = (size_t)(g * RT + rt) * 16 + (lane & 15);
                const float* gp = gi1 + (row * T + gi_t) * (3 * GRU_H) + 48 * J + 4 * q; // tile-major rows: [25 J][3 gates][16]
                const f32x4 z4 = *reinterpret_cast<const f32x4*>(gp);
                const f32x4 r4 = *reinterpret_cast<const f32x4*>(gp + 16);
                const f32x4 n4 = *reinterpret_cast<const f32x4*>(gp + 32);
                xci[((rt * 2 + tslot) * 3 + 0) * 64 + lane] = z4;
                xci[((rt * 2 + tslot) * 3 + 1) * 64 + lane] = r4;
                xci[((rt * 2 + tslot) * 3 + 2) * 64 + lane] = n4;
            }
        }
        __syncthreads();
    };

    // gate math + publish of step t: the helper wavefront of each tile, for every row tile of the group.
    // `first`: h_{t-1} = 0, so R h + Rb = Rb.  xci holds gi (layer 1) or W_ih h1_t (layer 2: Wb is added here).
    auto gates_and_publish = [&](int t, bool first) {
        if (helper) {
            for (int rt = 0; rt < my_rt; ++rt) {
                const int x0 = (rt * 2 + tslot) * 3 * 64 + lane;
                f32x4 giz, gir, gin;
                if (layer == 0 && rt < GPRE) {
                    giz = rt == 0 ? gpre[0][0] : gpre[1][0];
                    gir = rt == 0 ? gpre[0][1] : gpre[1][1];
                    gin = rt == 0 ? gpre[0][2] : gpre[1][2];
                } else {
                    giz = xci[x0] + wbz; gir = xci[x0 + 64] + wbr; gin = xci[x0 + 128] + wbn;
                }
                f32x4 az = (f32x4){0.f, 0.f, 0.f, 0.f}, ar = az, an = az;
                if (!first) { az = xch[x0]; ar = xch[x0 + 64]; an = xch[x0 + 128]; }
                const f32x4 hp = hpv[(rt * 2 + tslot) * 64 + lane];
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = fast_sigmoid(giz[r] + (az[r] + bz[r]));
                    const float rr = fast_sigmoid(gir[r] + (ar[r] + br[r]));
                    const float n = fast_tanh(gin[r] + rr * (an[r] + bn[r]));
                    h[r] = (1.0f - z) * n + z * hp[r];
                }
                hpv[(rt * 2 + tslot) * 64 + lane] = h;
                const int rtg = g * RT + rt;
                if (layer == 0) {
                    const unsigned off = (unsigned)((((t & 3) * n_rt + rtg) * GRU_J + J) * 1024);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, off, WS_AUX_SC1);
                } else {
                    const unsigned off = (unsigned)((((t & 1) * n_rt + rtg) * GRU_J + J) * 1024);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs2, lane16, off, WS_AUX_SC1);
                    const size_t row = (size_t)rtg * 16 + (lane & 15);
                    if (!WS_DIAG(variant, 2)) *reinterpret_cast<f32x4*>(hout2 + (row * T + t) * GRU_H + 16 * J + 4 * q) = h;
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // every storing wave drains before the flag is raised
        __syncthreads();
        if (tid == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };

    if (layer == 0) {
        // t = 0: gi only (fetched by the helpers; no product: h_{-1} = 0)
        request_gi(0);
        if (helper)
            for (int rt = GPRE; rt < my_rt; ++rt) {
                const size_t row = (size_t)(g * RT + rt) * 16 + (lane & 15);
                const float* gp = gi1 + (row * T) * (3 * GRU_H) + 48 * J + 4 * q;
                xci[((rt * 2 + tslot) * 3 + 0) * 64 + lane] = *reinterpret_cast<const f32x4*>(gp);
                xci[((rt * 2 + tslot) * 3 + 1) * 64 + lane] = *reinterpret_cast<const f32x4*>(gp + 16);
                xci[((rt * 2 + tslot) * 3 + 2) * 64 + lane] = *reinterpret_cast<const f32x4*>(gp + 32);
            }
        gates_and_publish(0, true);
        for (int t = 1; t < T; ++t) {
            // h1_{t-1} of every peer, and -- before slot t % 4 is overwritten -- h1_{t-4} consumed by every layer-2 peer
            // (layer 2 reads h1_s in its step s: it has published h2_{t-4}, flag t - 3, only after that)
            request_gi(t);
            if (!wait_for((unsigned)t, (t >= 4 && !WS_DIAG(variant, 4)) ? (unsigned)(t - 3) : 0u)) return;
            product(true, (unsigned)(((t - 1) & 3) * n_rt * GRU_J) * 1024u, false, 0u, t);
            gates_and_publish(t, false);
        }
    } else {
        // step t: gi2_t = W_ih h1_t (first half) beside R2 h2_{t-1} (second half), both from one fetch phase
        for (int t = 0; t < T; ++t) {
            const bool useA = !WS_DIAG(variant, 1);
            if (!wait_for(useA ? (unsigned)(t + 1) : 0u, (unsigned)t)) return;
            product(useA, (unsigned)((t & 3) * n_rt * GRU_J) * 1024u, t >= 1, (unsigned)(((t - 1) & 1) * n_rt * GRU_J) * 1024u, -1);
            gates_and_publish(t, t == 0);
        }
    }
}

// ------------------------------------------------------------------ the same, K split over 16 wavefronts
// gru_ws2_kernel keeps a gate's whole row of fragments (25 blocks) in one wavefront: six gate wavefronts per
// workgroup, two on each of three SIMDs, the fourth SIMD idle -- a step's MFMA phase is 200 MFMAs deep on a SIMD
// (tools/ws2_variants.py: 6.2 us per step for layer 1 alone against gru_ws_kernel's 5.4 with 100).  Here a workgroup
// has 16 wavefronts: twelve gate wavefronts (wave % 4 = SIMD: three on each), each holding ONE of the two
// accumulation chains of a gate -- the even or the odd super-steps, 13 or 12 fragment blocks, 52 VGPRs -- and four
// others (two tile helpers, two that poll flags and fetch operands).  The two chains of a gate meet in the gate math,
// a0 + a1 as before: the same bits as gru_ws2_kernel.  156 MFMAs per SIMD and step instead of 200.
// Also: one flag per unit tile, raised by the tile's helper wavefront itself right after its own drain (no workgroup
// barrier between publish and flag), and layer 2's row-major copy of h2 is stored after the flag.
// One row tile per group (16 wavefronts x 128 VGPRs leave no room for a second one's operands): up to 6 groups = 96
// sequences -- BASELINE config 3's 82 chunks, every live push; the 8-wavefront kernel above serves 2 to 4 row tiles
// per group.  What a step costs is memory round trips (flag, poll, fetch, drain: tools/ws2_trace.py), and what the polls
// cost depends on WHEN they are made: see `timed` in the kernel.
// waits before a step's first poll, in 10 ns ticks (see `timed` in the kernel): layer 1 / layer 2, groups of 13 + 25 and of 25 + 25
constexpr unsigned WS2K_WAIT_L1 = 150, WS2K_WAIT_L2 = 270, WS2K_WAIT_L1_ONE = 220, WS2K_WAIT_L2_ONE = 220;
// the same with layer 1's input projection in the kernel (GI1K): the projection's MFMAs run in the hand-off's shadow and move
// the moment the peers' flags become visible; swept again with tools/ws2_delay.py (82 sequences: 370 us at the waits above,
// 324 us at these; one chunk: 287 -> 280 us); layer 2's wait 3.2 -> 2.8 us after the context option ws2_calibrate had picked
// (240, 280) on every box it ran on (three in a row: 328 -> 322 us)
constexpr unsigned WS2K_WAIT_L1_G = 240, WS2K_WAIT_L2_G = 280, WS2K_WAIT_L1_ONE_G = 240, WS2K_WAIT_L2_ONE_G = 240;
// layer 2's fetch of the next h1 (wavefront 15) in groups of 13 + 25: 0.505 -> 0.491 ms at 82 chunks (0.55 ms at twice this
// wait: it is on the critical path then); none in groups of 25 + 25, where it costs a one-chunk push 10-25 us
constexpr unsigned WS2K_WAIT_H1 = 120;
// With layer 1 exchanging inside one XCD (lsync) its loop is ~1 us per step shorter and LAYER 2 paces the pipeline: the time no
// longer depends on layer 1's wait (0 .. 2.4 us: the same), and layer 2's waits move -- swept on a grid (layer 1 x layer 2 x
// layer 2's h1 wait; 82 sequences: 336 us without the local exchange, 289 - 292 us anywhere in 0 .. 0.4 / 1.6 .. 2.0 / 0.2 .. 0.5 us;
// 64 sequences: 301 -> 263 - 266 us at 1.4 .. 1.8 us and no h1 wait; one sequence: 288 -> 254 - 258 us at 1.2 .. 1.4 us)
// With layer 2 inside one XCD as well (at most four groups) an early poll is an L2 hit and costs nothing: the time is flat
// (228 - 233 us at 1, 32 and 64 sequences) for layer-2 waits of 0 .. 1.2 us and any layer-1 wait, 5 - 10 us more from 1.4 us on
constexpr unsigned WS2K_WAIT_L1_LOCAL = 20, WS2K_WAIT_L2_LOCAL = 180, WS2K_WAIT_L2_LOCAL_ONE = 140, WS2K_WAIT_L2_LOCAL_BOTH = 60, WS2K_WAIT_H1_LOCAL = 32;

// GI1K: layer 1 computes its input projection in the kernel (feat = the features, rows of kFeatStride floats); otherwise
// `feat` is gi1, the output of the GEMM launched in front (tile-major rows of 1200), as in round 3 -- kept for A/B runs on
// one box (context option ws2_variant bit 1024) and as the form the 8-wavefront kernel still uses
template <bool TRACE, bool GI1K, bool LOCAL>
__global__ __launch_bounds__(1024) void gru_ws2k_kernel(const float* __restrict__ feat, const float* __restrict__ W1frag,
                                                        const float* __restrict__ bG1, const float* __restrict__ R1frag,
                                                        const float* __restrict__ bR1, const float* __restrict__ W2frag,
                                                        const float* __restrict__ bW2, const float* __restrict__ R2frag,
                                                        const float* __restrict__ bR2, float* __restrict__ hout2,
                                                        float* hx1, float* hx2, unsigned* flags1, unsigned* flags2,
                                                        unsigned* err, int T, int n_rt, unsigned long long spin_ticks,
                                                        int variant, int nl1, unsigned waits, float* hx1l, unsigned* lsync)
{
    // lsync != nullptr: LAYER 1 EXCHANGES h1 INSIDE ONE XCD.  Layer 1 is the loop that paces the pipeline, and four memory round
    // trips of its hand-off cross the fabric (tools/handoff_rate.hip: 2.5 us per step for 13 workgroups spread over the chip,
    // 1.4 us when all of them sit on one XCD, store plainly -- the tile stays in that XCD's L2 -- and read with L2-served sc1
    // loads).  Workgroups are dealt to the XCCs round-robin (idle_cu_census.hip), so every XCC gets an eighth of the grid; a
    // workgroup reads the XCC it is on, takes a ticket of that XCC, and the first tickets of the first XCCs become the layer-1
    // sets (two groups of 13 per XCC, or one group of 25), everybody else takes a layer-2 place from one global ticket -- the
    // grouping is a fact read from the hardware, not an assumption about dispatch.  Layer 1 then publishes every tile twice: a
    // plain store + plain flag for its peers (the critical path), and behind that the write-through store + sc1 flag into the
    // ring layer 2 reads from wherever it is (with the ring's back-pressure checked there, off the critical path).
    //   lsync: [0, 160) layer 1's local flags, [160, 168) tickets per XCC, [168] the layer-2 ticket, [176, 336) layer 2's local flags
    //   (with at most four groups of 25 + 25 every layer-2 set has an XCC of its own as well) -- zeroed per pass like the flags.
    // dynamic LDS, in float4s: hbA[25][64], hbB[25][64] (the row tile's h1 and h2; layer 1 has no h2 and keeps the step's
    // input rows x_t there: two parities of 11 blocks); per tile slot: xch[3 gates][2 chains][64] recurrent partial products;
    // xci[2 parities][12 blocks][64]: the input projection of the step (layer 1: W' x_t for its two tiles; layer 2: W_ih h1,
    // 6 blocks per parity); hpv[64] the previous h of the tile; btab[2 tile slots][6][4] the biases (Rb z, r, n; the input
    // projection's bias z, r, n) as the four float4s a lane quartet needs; words; the step trace; wx[2 tiles][3 gates][11][64]:
    // layer 1's stationary input-projection fragments (W' = W_ih W_fc1, K = 176)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int O_HBB = GRU_J * 64, O_XCH = 2 * GRU_J * 64, O_XCI = O_XCH + 12 * 64, O_HPV = O_XCI + 24 * 64, O_BTAB = O_HPV + 2 * 64,
                  O_WORDS = O_BTAB + 48, O_TRACE = O_WORDS + 1, O_WX = O_WORDS + 176;
    f32x4* hbA = reinterpret_cast<f32x4*>(smem);
    f32x4* hbB = hbA + O_HBB;
    f32x4* xch = hbA + O_XCH;
    f32x4* xci = hbA + O_XCI;
    f32x4* hpv = hbA + O_HPV;
    f32x4* btab = hbA + O_BTAB;
    f32x4* wx = hbA + O_WX;
    // (LDS-space pointers, not generic ones, for everything that is not plain array indexing: the flat -> LDS casts of
    // generic pointers made the backend emit an illegal instruction in some variants of this kernel, ROCm 7.2)
    __attribute__((address_space(3))) char* lds3 = (__attribute__((address_space(3))) char*)smem; // LDS-DMA targets, words, trace
    __attribute__((address_space(3))) volatile int* s_dead = (__attribute__((address_space(3))) volatile int*)(lds3 + O_WORDS * 16);
    __attribute__((address_space(3))) volatile int* hA_ready = s_dead + 1; // layer 2: h1 of steps < *hA_ready is in hbA (set by wavefront 15)
    __attribute__((address_space(3))) volatile int* g_done = s_dead + 2;   // layer 1: [tile slot] steps whose gate math is done (set by the tile's helper)
    // variant & 64 (tools/ws2_trace.py): the first layer-1 and the first layer-2 workgroup of group 0 keep shader-clock
    // stamps of every step's events in LDS and copy them behind the polled words (flags1 + 520 ...) when they are done
    __attribute__((address_space(3))) unsigned* trl = (__attribute__((address_space(3))) unsigned*)(lds3 + O_TRACE * 16);
#define WS_STAMP(t_, k_) do { if (TRACE && tr && lane == 0) trl[(t_) * 12 + (k_)] = (unsigned)clock64(); } while (0)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;
    // nl1 = layer-1 workgroups per group: 13 (two unit tiles each; 38 workgroups per group, six groups on 256 CUs) or,
    // when the launch has few enough row tiles for 50 workgroups per group, 25 (one unit tile each: the layer-1 step's
    // MFMA phase, which paces the whole pipeline, is 104 MFMAs deep on a SIMD instead of 156)
    const int gsz = nl1 + GRU_J;
    const bool one = nl1 == GRU_J;
    int g_ = (int)blockIdx.x / gsz;               // the group = the row tile
    int r38_ = (int)blockIdx.x - g_ * gsz;
    if (LOCAL) {
        __attribute__((address_space(3))) volatile int* s_role = (__attribute__((address_space(3))) volatile int*)(lds3 + O_WORDS * 16) + 3;
        if (tid == 0) {
            unsigned xcc = 0;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            xcc &= 7u;
            const int tk = (int)atomicAdd(lsync + 160 + xcc, 1u);
            const int per_host = one ? GRU_J : 26;                         // layer-1 workgroups an XCC hosts: one group of 25, or two of 13
            const int gl = one ? (int)xcc : 2 * (int)xcc + tk / 13;        // the layer-1 group this ticket belongs to
            int role;
            if (tk < per_host && gl < n_rt) role = gl * 64 + (one ? tk : tk % 13);
            else if (one && 2 * n_rt <= 8) // every set has an XCC of its own: XCC n_rt + g hosts layer 2 of group g
                role = ((int)xcc >= n_rt && (int)xcc < 2 * n_rt && tk < GRU_J) ? ((int)xcc - n_rt) * 64 + nl1 + tk : -1;
            else {
                const int t2 = (int)atomicAdd(lsync + 168, 1u);
                role = t2 < n_rt * GRU_J ? (t2 / GRU_J) * 64 + nl1 + t2 % GRU_J : -1; // -1: the launch has more workgroups than places
            }
            s_role[0] = role;
        }
        __syncthreads();
        const int role = s_role[0];
        if (role < 0) return;
        g_ = role >> 6;
        r38_ = role & 63;
    }
    const int g = g_, r38 = r38_;
    const int layer = r38 >= nl1;
    const bool loc = LOCAL && !layer;  // this workgroup exchanges h1 through its XCD's L2
    // ... and with at most four groups of 25 + 25 every layer-2 set sits on an XCC of its own too: h2 goes the same way
    const bool loc2 = LOCAL && layer && one && 2 * n_rt <= 8;
    const int pair = layer ? r38 - nl1 : r38;     // layer 1: pair of unit tiles (or the unit tile); layer 2: the unit tile
    const bool tr = TRACE && g == 0 && (r38 == 0 || r38 == nl1);
    const bool gate_wave = wave < 12;
    const int ws = gate_wave ? wave / 6 : (wave == 13); // half: layer 1 the tile of the pair, layer 2 the matrix (W_ih / R2)
    const int gk = wave % 6;
    const int wg = gk % 3;                         // gate
    const int kp = gk / 3;                         // accumulation chain: even (0) or odd (1) super-steps
    const int J = (layer || one) ? pair : 2 * pair + ws;
    const bool tile_ok = (!layer && one) ? ws == 0 : J < GRU_J; // layer 1: the 13th workgroup's second tile does not exist; one tile per workgroup: the second half idles
    const int Jc = tile_ok ? J : 0;
    const int tslot = layer ? 0 : ws;
    // who does the gate math, fetches gi and publishes: layer 1: wavefronts 12 / 13 for the two tiles; layer 2: 12
    const bool helper = layer ? wave == 12 : ((wave == 12 || wave == 13) && tile_ok);
    // Everything on a step's critical path outside the product phase -- gate math, drain, flag, poll, fetch -- is done by
    // wavefronts 12..15, the youngest on their SIMDs: without priority their instructions queue behind whatever the three gate
    // wavefronts there issue (the input projections run in the hand-off's shadow, i.e. exactly then)
    if (wave >= 12) __builtin_amdgcn_s_setprio(3);

    if (tid == 0) { *s_dead = (int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *hA_ready = 0; g_done[0] = 0; g_done[1] = 0; }
    if (tid < 2 * 64) hpv[tid] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int i = tid; i < 24 * 64; i += 1024) xci[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; // (layer 1 with GI1K: the chain-1 slots are never written)
    if (tid >= 128 && tid < 128 + 48) { // biases of the workgroup's tile(s): read from LDS in the gate math
        const int e = tid - 128, sl = e / 24, k = (e % 24) / 4, qq = e & 3;
        const int Jb = (layer || one) ? pair : 2 * pair + sl;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (Jb < GRU_J && (layer ? sl == 0 : ((GI1K || k < 3) && (!one || sl == 0)))) {
            // Rb and layer 2's Wb are gate-major [3][400]; layer 1's folded bias b' = W_ih b_fc1 + Wb comes tile-major [25][3][16]
            const float* src = k < 3 ? (layer ? bR2 : bR1) + (k % 3) * GRU_H + 16 * Jb + 4 * qq
                                     : (layer ? bW2 + (k % 3) * GRU_H + 16 * Jb + 4 * qq : bG1 + 48 * Jb + 16 * (k % 3) + 4 * qq);
            v = *reinterpret_cast<const f32x4*>(src);
        }
        btab[e] = v;
    }
    __syncthreads();
    if (*s_dead) return;
    if (WS_DIAG(variant, 4) && layer) return;
    __syncthreads();

    // ---- stationary weights of a gate wavefront: chain kp of gate wg of tile J: R1 (layer 1), or layer 2's W_ih
    // (first half) / R2 (second half)
    constexpr int NW = (GRU_J + 1) / 2;
    f32x4 w[NW];
    {
        const bool gate = gate_wave && tile_ok;
        const size_t blk = ((size_t)(Jc * 3 + wg) * GRU_J) * 64 + lane;
        const f32x4* src = reinterpret_cast<const f32x4*>(layer ? (ws ? R2frag : W2frag) : R1frag) + blk;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int S = 2 * i + kp;
            w[i] = (gate && S < GRU_J) ? src[(S < GRU_J ? S : 0) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // Layer 1 computes its own input projection gi1_t = W' x_t + b' (W' = W_ih W_fc1: fc1 folded, K = 161 -> 11 super-steps),
    // like layer 2 does with W_ih h1: the fragments of its tile(s) are stationary in LDS (wx), the step's input rows -- the
    // log-power features of the group's 16 sequences, 11 KB -- are fetched a step ahead by wavefront 15 straight into LDS, and
    // the gate wavefronts run the 20-24 MFMAs of a chain in the shadow of the h1 hand-off, where they have nothing else to do.
    // (32-bit lane offsets into buffer resources, the step as the scalar offset: the 128-VGPR budget has no room for
    // 64-bit per-lane addresses)
    const auto rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat), 0, n_rt * 16 * T * kFeatStride * 4, 0x00020000);
    const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(hout2, 0, n_rt * 16 * T * GRU_H * 4, 0x00020000);
    const unsigned out_base = (unsigned)((g * 16 * T) * GRU_H + 16 * Jc) * 4u;
    // (not GI1K) layer 1's gi of the step comes from the GEMM in front: requested BEFORE the wait
    const auto rs_gi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(feat), 0, n_rt * 16 * T * (3 * GRU_H) * 4, 0x00020000);
    const unsigned gi_lane = (unsigned)(((g * 16 + (lane & 15)) * T) * (3 * GRU_H) + 48 * Jc + 4 * q) * 4u; // tile-major rows: [25 J][3 gates][16]
    f32x4 gpre[3];
    auto request_gi = [&](int t) {
        if (!GI1K && layer == 0 && helper) {
            const unsigned so = (unsigned)t * (3 * GRU_H * 4);
            gpre[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gi_lane, so, 0));
            gpre[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gi_lane + 64u, so, 0));
            gpre[2] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gi_lane + 128u, so, 0));
        }
    };
    constexpr int XS = kFeatStride / 16; // 11 super-steps
    // wavefront 15 of a layer-1 workgroup: rows t of the group's sequences -> hbB[parity of t], in MFMA operand layout
    // (block S: lane (m, q) <- feat[sequence m][t][16 S + 4 q .. + 3])
    auto stage_x = [&](int t) {
        if (WS_DIAG(variant, 256)) return; // timing only: no input rows (garbage in, garbage out)
        unsigned xl = (unsigned)(lane & 15);
        asm volatile("" : "+v"(xl)); // the lane's offset is formed here, per step: hoisted, it would be spilled
        const unsigned x_lane = ((unsigned)(g * 16) + xl) * (unsigned)(T * kFeatStride * 4) + (unsigned)q * 16u;
#pragma unroll
        for (int S = 0; S < XS; ++S)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (__attribute__((address_space(3))) void*)(lds3 + (O_HBB + ((t & 1) * XS + S) * 64) * 16), 16,
                                                     x_lane, (unsigned)t * (kFeatStride * 4) + S * 64, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };

    // ONE buffer resource over the whole exchange area (the launcher lays it out contiguously: four ring slots of h1, two slots of
    // h2, two XCD-local slots of h1): three resources were twelve scalar registers of a kernel that has none to spare
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(hx1, 0, 10 * n_rt * GRU_J * 1024, 0x00020000);
    const unsigned o_h2 = (unsigned)((loc2 ? 8 : 4) * n_rt * GRU_J) * 1024u, o_loc = (unsigned)(6 * n_rt * GRU_J) * 1024u; // (layer 2's XCD-local slots: 8 .. 10)
    (void)hx2; (void)hx1l;
    const unsigned lane16 = (unsigned)lane * 16u;
    // one flag per unit tile and layer (25 + 25 per group)
    __attribute__((address_space(1))) unsigned* my_flag =
        (__attribute__((address_space(1))) unsigned*)((layer ? flags2 : flags1) + g * GRU_J + Jc);
    // the polling wavefront (14): lanes 0..24 watch layer 1's flags of this group, lanes 32..56 layer 2's
    __attribute__((address_space(1))) unsigned* poll_flag =
        (__attribute__((address_space(1))) unsigned*)(lane < 32 ? (loc ? lsync : flags1) + g * GRU_J + (lane < GRU_J ? lane : 0)
                                                                : (loc2 ? lsync + 176 : flags2) + g * GRU_J + (lane - 32 < GRU_J ? lane - 32 : 0));
    __attribute__((address_space(1))) unsigned* my_flag_l = (__attribute__((address_space(1))) unsigned*)((LOCAL ? lsync + (layer ? 176 : 0) : flags1) + g * GRU_J + Jc);

    // (Hand-off form: the guide's write-through one -- sc1 payload, drain, sc1 flag; the polling wavefront loads only after
    // its poll has matched -- with sc1 LDS-DMA loads where the guide's measured table has sc1 loads to registers, and one
    // flag per storing wavefront; tools/ws_stress.py is the evidence for both: DESIGN.md section 3.1.)
    // The polling wavefront (14) does the whole acquisition of a step's operand: it waits until layer 1 has published
    // `need1` steps and layer 2 `need2` (0 = no requirement), then fetches the row tile's h (25 KB, 25 blocks) straight
    // into LDS -- sc1 LDS-DMA loads, no registers, no second barrier -- and drains them; the other wavefronts meet it
    // at the workgroup barrier.  On a deadline it raises the error word instead.  Uniform false on a deadline.
    const unsigned row0 = (unsigned)(g * GRU_J) * 1024u;
    // pw: the wavefront that does it (14; layer 2 also uses 15 for h1, so that the two acquisitions of a step run side by
    // side); ready >= 0: afterwards *hA_ready = ready (the wavefronts that consume hbA wait for that word, not for a barrier)
    // timed: wait before the first poll.  A poll of flags that cannot be up yet (the peers publish ~1 us after the barrier
    // this follows, and a flag takes ~0.5 us more to become visible) is a wasted ~0.9 us round trip AND traffic on the
    // group's flag lines, which slows every store and fetch of the step: the right wait took 70 us off the 350 us of a
    // one-chunk launch and 55 off the 400 of 82 chunks (tools/ws2_delay.py).  The wait is a wall-clock interval from the
    // barrier (s_memrealtime, 100 MHz), per layer and group shape, a little short of the measured optimum: too short only
    // brings the wasted polls back, too long would sit on the critical path.  (A self-tuning wait -- longer after a step
    // that needed several polls, shorter after a first-poll hit -- ratchets up across the group, because a late poller
    // publishes late and makes its peers' polls miss: 400 us instead of 350.)  Timing only: results do not depend on it.
    // waits != 0 (context options ws2_waits / ws2_calibrate): layer 1's wait in the low half, layer 2's in the high half, ticks.
    const unsigned wait_ticks = waits ? (layer ? waits >> 16 : waits & 0xFFFFu)
                                      : LOCAL ? (layer ? (one ? (loc2 ? WS2K_WAIT_L2_LOCAL_BOTH : WS2K_WAIT_L2_LOCAL_ONE) : WS2K_WAIT_L2_LOCAL) : WS2K_WAIT_L1_LOCAL)
                                                      : GI1K ? (layer ? (one ? WS2K_WAIT_L2_ONE_G : WS2K_WAIT_L2_G) : (one ? WS2K_WAIT_L1_ONE_G : WS2K_WAIT_L1_G))
                                                             : (layer ? (one ? WS2K_WAIT_L2_ONE : WS2K_WAIT_L2) : (one ? WS2K_WAIT_L1_ONE : WS2K_WAIT_L1));
    auto acquire = [&](int pw, unsigned need1, unsigned need2, bool from_h2, unsigned slot, int st, int ev, int ready, bool timed) {
        if (wave == pw) {
            const unsigned wt = timed ? wait_ticks : ((pw == 15 && !one) ? (LOCAL ? WS2K_WAIT_H1_LOCAL : WS2K_WAIT_H1) : 0u);
            if (wt) {
                const unsigned long long until = __builtin_amdgcn_s_memrealtime() + wt;
                while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(1);
            }
            WS_STAMP(st, ev);
            const unsigned need = lane < 32 ? need1 : need2;
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            int dead = 0;
            // (one load at a time: a flag load takes ~1800 clocks to come back, but four of them in flight ~450 clocks apart
            // made every step SLOWER -- 15.8k clocks against 14.8k: the extra sc1 traffic to the groups' flag lines delays
            // the fetches and the drains more than the earlier notice saves; tools/ws2_trace.py)
            // (fetching each tile as its flag arrives instead of all 25 behind the last flag was measured and dropped: -6 us of
            // 420 at 82 sequences, +14...+20 us at one chunk -- the group's flags arrive together, there is no skew to hide behind)
            for (;;) {
                unsigned v = 0xFFFFFFFFu; // only the lanes that wait for something load: one flag line per poll, not two
                if (need) v = __hip_atomic_load(poll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (__all(v >= need)) break;
                __builtin_amdgcn_s_sleep(1);
                if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) { dead = 1; break; }
            }
            WS_STAMP(st, ev + 1);
            if (!dead) {
                if (from_h2) {
#pragma unroll
                    for (int S = 0; S < GRU_J; ++S)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(lds3 + (GRU_J + S) * 1024), 16, lane16,
                                                                 o_h2 + slot + row0 + S * 1024, 0, WS_AUX_SC1);
                } else if (loc) {
#pragma unroll
                    for (int S = 0; S < GRU_J; ++S)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(lds3 + S * 1024), 16, lane16,
                                                                 o_loc + slot + row0 + S * 1024, 0, WS_AUX_SC1);
                } else {
#pragma unroll
                    for (int S = 0; S < GRU_J; ++S)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(lds3 + S * 1024), 16, lane16,
                                                                 slot + row0 + S * 1024, 0, WS_AUX_SC1);
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            WS_STAMP(st, ev + 2);
            if (lane == 0 && dead) {
                __hip_atomic_store((__attribute__((address_space(1))) unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_dead = 1;
            }
            if (lane == 0 && !dead && ready >= 0) *hA_ready = ready; // behind the drain: the bytes are in LDS
        }
    };
    // a consumer of hbA waits for the word (or for the error word); uniform
    auto wait_hA = [&](int need) -> bool {
        for (;;) {
            if (*hA_ready >= need) break;
            if (*s_dead) return false;
            __builtin_amdgcn_s_sleep(1);
        }
        asm volatile("" ::: "memory");
        return true;
    };
    auto barrier_alive = [&]() -> bool {
        __syncthreads();
        return *s_dead == 0; // uniform: every wavefront reads the same word after the barrier
    };

    // one chain of one gate of one matrix: this gate wavefront's fragments x the row tile's h in LDS -> its slot of
    // xch (layer 1: R1 h1; layer 2: R2 h2, second half) or xci (layer 2: W_ih h1, first half; xci follows xch)
    const unsigned xoff_s = (unsigned)__builtin_amdgcn_readfirstlane((((layer && !ws) ? 12 : 0) + (tslot * 3 + wg) * 2 + kp) * 1024);
    auto chain = [&](const f32x4* hsrc, unsigned xadd) {
        const f32x4* hb = hsrc + kp * 64 + lane; // chain kp: super-steps kp, kp + 2, ...
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (i < NW - 1 || kp == 0) { // the odd chain has 12 blocks, the even one 13
                const f32x4 hv = hb[i * 128];
                a = MFMA16(w[i].x, hv.x, a);
                a = MFMA16(w[i].y, hv.y, a);
                a = MFMA16(w[i].z, hv.z, a);
                a = MFMA16(w[i].w, hv.w, a);
            }
        }
        unsigned xoff = xoff_s + xadd;
        if (!TRACE) asm volatile("" : "+s"(xoff)); // the per-lane address is formed here, per step: hoisted, it would be spilled
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xch) + xoff + lane16) = a;
    };

    // layer 1's input projection of step t: gate wg of this wavefront's tile, fragments from wx, rows from hbB[parity of t] ->
    // its slot of xci[parity of t].  ONE chain over the 11 super-steps, from zero, the bias added behind it in the gate math:
    // the accumulation order of the GEMM this replaces (panel_gemm_s_kernel), so a launch has the same bits whether its
    // input projection ran here or in a GEMM in front -- the wavefronts of chain 0 do it, those of chain 1 leave their slot
    // of xci zero (44 dependent MFMAs = 1.8k clocks, in a shadow of ~7k)
    auto chain_x = [&](int t) {
        if (WS_DIAG(variant, 512)) return; // timing only: no input projection
        if (kp) return;
        const f32x4* wb = wx + ((tslot * 3 + wg) * XS) * 64 + lane;
        const f32x4* xb = hbB + ((t & 1) * XS) * 64 + lane;
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < XS; ++i) {
            const f32x4 wv = wb[i * 64], xv = xb[i * 64];
            a = MFMA16(wv.x, xv.x, a);
            a = MFMA16(wv.y, xv.y, a);
            a = MFMA16(wv.z, xv.z, a);
            a = MFMA16(wv.w, xv.w, a);
        }
        unsigned xoff = (unsigned)__builtin_amdgcn_readfirstlane((12 + (t & 1) * 12 + (tslot * 3 + wg) * 2) * 1024);
        if (!TRACE) asm volatile("" : "+s"(xoff));
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xch) + xoff + lane16) = a;
    };

    // gate math + publish of step t: the helper wavefront of each tile.  `first`: h_{t-1} = 0, so R h + Rb = Rb.
    // The input projection comes from xci (+ its bias): W' x_t (layer 1) or W_ih h1_t (layer 2).
    // xi0: where this tile's input projection of this step sits in xci (the two halves of xci alternate by step parity)
    auto gates_and_publish = [&](int t, bool first, int xi0) {
        if (helper) {
            if (wave == 12) WS_STAMP(t, 6);
            const f32x4* bt = btab + tslot * 24 + q;
            const int x0 = tslot * 6 * 64 + lane;
            const int xi = xi0 + lane;
            f32x4 z4, r4, h;
            {
                f32x4 gi = (GI1K || layer) ? (xci[xi] + xci[xi + 64]) + bt[12] : gpre[0];
                f32x4 a = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : xch[x0] + xch[x0 + 64];
                const f32x4 b = bt[0];
#pragma unroll
                for (int r = 0; r < 4; ++r) z4[r] = fast_sigmoid(gi[r] + (a[r] + b[r]));
            }
            {
                f32x4 gi = (GI1K || layer) ? (xci[xi + 128] + xci[xi + 192]) + bt[16] : gpre[1];
                f32x4 a = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : xch[x0 + 128] + xch[x0 + 192];
                const f32x4 b = bt[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) r4[r] = fast_sigmoid(gi[r] + (a[r] + b[r]));
            }
            {
                f32x4 gi = (GI1K || layer) ? (xci[xi + 256] + xci[xi + 320]) + bt[20] : gpre[2];
                f32x4 a = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : xch[x0 + 256] + xch[x0 + 320];
                const f32x4 b = bt[8];
                const f32x4 hp = hpv[tslot * 64 + lane];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float n = fast_tanh(gi[r] + r4[r] * (a[r] + b[r]));
                    h[r] = (1.0f - z4[r]) * n + z4[r] * hp[r];
                }
            }
            hpv[tslot * 64 + lane] = h;
            if (loc) {
                // for the peers, inside the XCD: a plain store (the line stays in this XCD's L2), the drain, a plain flag
                const unsigned offl = (unsigned)((((t & 1) * n_rt + g) * GRU_J + J) * 1024);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, o_loc + offl, 0);
                if (wave == 12) WS_STAMP(t, 7);
                if (GI1K && lane == 0) g_done[tslot] = t + 1;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (wave == 12) WS_STAMP(t, 8);
                if (lane == 0) __hip_atomic_store(my_flag_l, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                // for layer 2, wherever it runs: the ring of four slots, behind its back-pressure (layer 2 has published step
                // t - 4 before slot t % 4 is overwritten: it reads h1_s in its step s) -- all of it off layer 1's critical path
                if (t >= 4 && !WS_DIAG(variant, 4)) {
                    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
                    __attribute__((address_space(1))) unsigned* f2 = (__attribute__((address_space(1))) unsigned*)(flags2 + g * GRU_J + (lane < GRU_J ? lane : 0));
                    for (;;) {
                        const unsigned v = lane < GRU_J ? __hip_atomic_load(f2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0xFFFFFFFFu;
                        if (__all(v >= (unsigned)(t - 3))) break;
                        if (*s_dead) break;
                        if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) {
                            if (lane == 0) { __hip_atomic_store((__attribute__((address_space(1))) unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_dead = 1; }
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                const unsigned off = (unsigned)((((t & 3) * n_rt + g) * GRU_J + J) * 1024);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, off, WS_AUX_SC1);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                return;
            }
            if (layer == 0) {
                const unsigned off = (unsigned)((((t & 3) * n_rt + g) * GRU_J + J) * 1024);
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, off, WS_AUX_SC1);
            } else {
                const unsigned off = (unsigned)((((t & 1) * n_rt + g) * GRU_J + J) * 1024);
                if (loc2) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, o_h2 + off, 0);
                else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, o_h2 + off, WS_AUX_SC1);
            }
            // this wavefront stored the whole tile: it drains and raises the tile's flag itself
            if (wave == 12) WS_STAMP(t, 7);
            if (GI1K && layer == 0 && lane == 0) g_done[tslot] = t + 1; // the gate wavefronts may start the next step's input projection
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (wave == 12) WS_STAMP(t, 8);
            // (layer 2 inside one XCD: the plain flag its peers poll first, then the sc1 flag layer 1's back-pressure check reads
            // across the chip -- that one stands for "h1 of this step has been consumed", not for any stored byte)
            if (loc2 && lane == 0) __hip_atomic_store(my_flag_l, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (lane == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (layer && !WS_DIAG(variant, 2)) { // the row-major copy the next layer (fc2) reads: not part of the hand-off
                unsigned l16 = lane16;
                asm volatile("" : "+v"(l16)); // the lane's offset is formed here, per step: hoisted, it would be spilled
                const unsigned out_lane = ((l16 >> 4) & 15u) * (unsigned)(T * GRU_H * 4) + (l16 >> 8) * 16u;
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs_out, out_lane, out_base + (unsigned)t * (GRU_H * 4), 0);
            }
        }
    };

    if (layer == 0) {
        if (GI1K) {
            // the stationary input-projection fragments of this workgroup's tile(s): 33 or 66 blocks of 1 KB, all wavefronts
            const int n_blk = (one ? 1 : 2) * 3 * XS;
            for (int b = wave; b < n_blk; b += 16) {
                const int sl = b / (3 * XS), r = b - sl * 3 * XS;
                const int Jb = one ? pair : 2 * pair + sl;
                f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
                if (Jb < GRU_J) v = reinterpret_cast<const f32x4*>(W1frag)[((size_t)Jb * 3 * XS + r) * 64 + lane];
                wx[b * 64 + lane] = v;
            }
            if (wave == 15) { stage_x(0); if (T > 1) stage_x(1); }
            __syncthreads();
            if (gate_wave && tile_ok) { chain_x(0); if (T > 1) chain_x(1); }
            __syncthreads();
        }
        request_gi(0);
        gates_and_publish(0, true, tslot * 6 * 64); // t = 0: the input projection only (no recurrent product: h_{-1} = 0)
        for (int t = 1; t < T; ++t) {
            // h1_{t-1} of every peer, and -- before slot t % 4 is overwritten -- h1_{t-4} consumed by every layer-2 peer
            // (layer 2 reads h1_s in its step s: it has published h2_{t-4}, flag t - 3, only after that)
            request_gi(t);
            if (loc) acquire(14, (unsigned)t, 0u, false, (unsigned)(((t - 1) & 1) * n_rt * GRU_J) * 1024u, t, 0, -1, true);
            else acquire(14, (unsigned)t, (t >= 4 && !WS_DIAG(variant, 4)) ? (unsigned)(t - 3) : 0u, false, (unsigned)(((t - 1) & 3) * n_rt * GRU_J) * 1024u, t, 0, -1, true);
            if (!barrier_alive()) return;
            if (wave == 0) WS_STAMP(t, 4);
            if (gate_wave && tile_ok) chain(hbA, 0u);
            // the product phase is when the group exchanges nothing: the next step's input rows are fetched here (chain_x(t) is
            // behind the barrier above: nobody reads either half of hbB now)
            if (GI1K && wave == 15 && t + 1 < T) stage_x(t + 1);
            if (wave == 0) WS_STAMP(t, 5);
            __syncthreads();
            gates_and_publish(t, false, (t & 1) * 12 * 64 + tslot * 6 * 64);
            // The next step's input projection (the other half of xci) belongs in the hand-off's shadow -- drain, flag, the peers'
            // polls: ~7k clocks in which the gate wavefronts have nothing to do -- and NOT beside the helper's gate math: started
            // right behind the barrier its MFMAs and LDS reads took the gate math from 0.8k to 3.2k clocks, 1.5 us per step
            // (tools/ws2_trace.py).  So the gate wavefronts wait for the word their tile's helper sets once h_t is stored.
            if (GI1K && gate_wave && tile_ok && t + 1 < T) {
                while (g_done[tslot] < t + 1 && *s_dead == 0) __builtin_amdgcn_s_sleep(2);
                chain_x(t + 1);
            }
        }
    } else {
        // Step t.  Right behind the barrier that ends step t - 1's R phase three things start side by side: the helper
        // does that step's gates and publishes; wavefront 15 fetches h1_t (layer 1 runs ahead through its ring of four
        // slots) and sets *hA_ready; wavefront 14 waits for the peers' h2_{t-1} and fetches it.  The first half's chains
        // compute gi2_t = W_ih h1_t as soon as hA_ready says so -- no barrier: they wait for the LDS word, and the two
        // halves of xci alternate so that they may overwrite nothing the helper still reads -- i.e. in the shadow of the
        // h2 hand-off; then ONE barrier (h2 in LDS, W_ih chains done, helper back), R2 h2_{t-1} on the second half's
        // chains, the barrier that ends the step.  On the chain from one publish to the next: flag, fetch of h2, ONE
        // product phase of 104 MFMAs per SIMD, gates, drain.
        const bool useA = !WS_DIAG(variant, 1);
        if (useA) acquire(15, 1u, 0u, false, 0u, 0, 0, 1, false);
        for (int t = 0; t < T; ++t) {
            const int par = t & 1;
            if (gate_wave && ws == 0 && useA && !WS_DIAG(variant, 32)) {
                if (wait_hA(t + 1)) chain(hbA, (unsigned)par * 6144u);
            }
            if (wave == 0) WS_STAMP(t, 3);
            if (!barrier_alive()) return;
            if (wave == 6) WS_STAMP(t, 4);
            if (gate_wave && ws == 1 && t >= 1) chain(hbB, 0u);
            if (wave == 6) WS_STAMP(t, 5);
            __syncthreads();
            if (t + 1 < T) {
                if (useA) acquire(15, (unsigned)(t + 2), 0u, false, (unsigned)(((t + 1) & 3) * n_rt * GRU_J) * 1024u, t, 9, t + 2, false);
                acquire(14, 0u, (unsigned)(t + 1), true, (unsigned)((t & 1) * n_rt * GRU_J) * 1024u, t + 1, 0, -1, true);
            }
            gates_and_publish(t, t == 0, par * 6 * 64);
        }
    }
    if (TRACE && tr) {
        __syncthreads();
        for (int i = tid; i < T * 12 && i < 1000; i += 1024) flags1[520 + layer * 1000 + i] = trl[i];
    }
#undef WS_STAMP
}

// ------------------------------------------------------------------ the same for several row tiles per group
// gru_ws2k_kernel serves one row tile per group (up to 96 sequences).  From there to ~1500 sequences a group of 13 + 25
// workgroups takes RT row tiles through every step, and what a step costs is no longer the hand-off alone but RT products:
// the 8-wavefront gru_ws2_kernel spends 9.4k clocks per row tile (six gate wavefronts on three SIMDs, operands staged through
// registers by all wavefronts, a barrier pair per row tile), 3x the 3.2k its MFMAs need.  This kernel keeps gru_ws2k's
// wavefront layout -- twelve gate wavefronts, one accumulation chain each (the same two chains per gate: the same bits as
// gru_ws2_kernel), three per SIMD; helpers 12 / 13; wavefronts 14 / 15 fetch -- and streams the row tiles through a
// double-buffered operand area: while the gate wavefronts multiply row tile rt, wavefront 14 (h1) and 15 (h2) fetch row tile
// rt + 1 straight into the other buffer (sc1 LDS-DMA, no registers) and the helper does row tile rt - 1's gate math and
// publishes its tile; ONE barrier per row tile.  The flags are polled once per step (all RT row tiles of a step are
// published before a flag is raised).  Layer 1's gi comes from the GEMM in front (tile-major rows), two row tiles ahead in
// the helper's registers.  Hand-off form, bounded spins, error word, fallback launch: gru_ws2k_kernel's.
// 10 ns ticks before a step's first poll.  Swept through the context option ws2_waits (its low half reaches this kernel): the pipelined
// recurrence at 256 / 1024 / 1536 chunks 717 / 1893 / 2627 us at 0.5 us, 679 / 1853 / 2617 at 1.0, 657 / 1841 / 2553 at 1.5 (rounds 4-5),
// 639 / 1823 / 2572 at 2.0, 655 / 1826 / 2586 at 2.5, 709 / 1862 / 2612 at 3.5: timing only, the digests are the same
constexpr unsigned WS2M_WAIT = 200;
__global__ __launch_bounds__(1024) void gru_ws2m_kernel(const float* __restrict__ gi1, const float* __restrict__ R1frag,
                                                        const float* __restrict__ bR1, const float* __restrict__ W2frag,
                                                        const float* __restrict__ bW2, const float* __restrict__ R2frag,
                                                        const float* __restrict__ bR2, float* __restrict__ hout2,
                                                        float* hx1, float* hx2, unsigned* flags1, unsigned* flags2,
                                                        unsigned* err, int T, int RT, int n_rt, unsigned long long spin_ticks, int variant,
                                                        unsigned first_poll_wait)
{
    // first_poll_wait: 10 ns ticks before a step's first poll (the launcher passes WS2M_WAIT, or the context option ws2_waits' low half)
    // diagnostics build only (timing, wrong results): variant & 4096 no operand fetch behind a step's first row tile, 8192 no gate
    // math (zeros are published), 16384 no products
    // dynamic LDS, in float4s: hb[2 buffers][A: 25 blocks of h1 | B: 25 blocks of h2][64]; xch[2 buffers][12 blocks][64]: a row
    // tile's partial products (layer 1: [tile slot][gate][chain]; layer 2: R2 h2 in blocks 0..5, W_ih h1 in 6..11);
    // hpv[RT][2 tile slots][64]: the previous h of the workgroup's tile(s), per row tile; btab[2][6][4]; words
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int O_XCH = 4 * GRU_J * 64, O_HPV = O_XCH + 24 * 64;
    f32x4* hb = reinterpret_cast<f32x4*>(smem);
    f32x4* xch = hb + O_XCH;
    f32x4* hpv = hb + O_HPV;
    f32x4* btab = hpv + (size_t)RT * 128;
    __attribute__((address_space(3))) char* lds3 = (__attribute__((address_space(3))) char*)smem;
    __attribute__((address_space(3))) volatile int* s_dead = (__attribute__((address_space(3))) volatile int*)(lds3 + (O_HPV + RT * 128 + 48) * 16);
    __attribute__((address_space(3))) volatile int* s_go = s_dead + 1; // layer 2: steps whose flags wavefront 14 has seen (wavefront 15 fetches h2 behind it)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4;
    const int g = blockIdx.x / 38;
    const int r38 = blockIdx.x - g * 38;
    const int layer = r38 >= 13;
    const int pair = layer ? r38 - 13 : r38;
    const bool gate_wave = wave < 12;
    const int ws = gate_wave ? wave / 6 : (wave == 13);
    const int gk = wave % 6;
    const int wg = gk % 3;
    const int kp = gk / 3;
    const int J = layer ? pair : 2 * pair + ws;
    const bool tile_ok = J < GRU_J;
    const int Jc = tile_ok ? J : 0;
    const int tslot = layer ? 0 : ws;
    const bool helper = layer ? wave == 12 : ((wave == 12 || wave == 13) && tile_ok);
    if (wave >= 12) __builtin_amdgcn_s_setprio(3); // helpers, poller, fetchers: few instructions, all of them on the critical path (no measurable effect here, unlike in gru_ws2k)
    const int my_rt = (n_rt - g * RT < RT) ? n_rt - g * RT : RT;

    if (tid == 0) { *s_dead = (int)__hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); *s_go = 0; }
    for (int i = tid; i < RT * 128; i += 1024) hpv[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tid >= 128 && tid < 128 + 48) {
        const int e = tid - 128, sl = e / 24, k = (e % 24) / 4, qq = e & 3;
        const int Jb = layer ? pair : 2 * pair + sl;
        f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
        if (Jb < GRU_J && (layer ? sl == 0 : k < 3)) {
            const float* src = (k < 3 ? (layer ? bR2 : bR1) : bW2) + (k % 3) * GRU_H + 16 * Jb + 4 * qq;
            v = *reinterpret_cast<const f32x4*>(src);
        }
        btab[e] = v;
    }
    __syncthreads();
    if (*s_dead) return;
    __syncthreads();

    // stationary weights: chain kp of gate wg of tile J: R1 (layer 1), layer 2's W_ih (first half) / R2 (second half)
    constexpr int NW = (GRU_J + 1) / 2;
    f32x4 w[NW];
    {
        const bool gate = gate_wave && tile_ok;
        const size_t blk = ((size_t)(Jc * 3 + wg) * GRU_J) * 64 + lane;
        const f32x4* src = reinterpret_cast<const f32x4*>(layer ? (ws ? R2frag : W2frag) : R1frag) + blk;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int S = 2 * i + kp;
            w[i] = (gate && S < GRU_J) ? src[(S < GRU_J ? S : 0) * 64] : (f32x4){0.f, 0.f, 0.f, 0.f};
        }
    }
    // (buffer descriptors are formed where they are used: four of them held across the loops would be 16 spilled SGPRs)
    const unsigned lane16 = (unsigned)lane * 16u;
    __attribute__((address_space(1))) unsigned* my_flag = (__attribute__((address_space(1))) unsigned*)((layer ? flags2 : flags1) + g * GRU_J + Jc);
    __attribute__((address_space(1))) unsigned* poll_flag =
        (__attribute__((address_space(1))) unsigned*)(lane < 32 ? flags1 + g * GRU_J + (lane < GRU_J ? lane : 0)
                                                                : flags2 + g * GRU_J + (lane - 32 < GRU_J ? lane - 32 : 0));

    // layer 1's gi of (row tile rt, step t) for this helper's tile: tile-major rows [25 J][3 gates][16]
    // (one row tile ahead: requested right after the gate math of the row tile before has consumed the registers, used a
    // whole product phase later)
    f32x4 gpre[3];
    auto request_gi = [&](int rt, int t) {
        const auto rs_gi = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gi1), 0, n_rt * 16 * T * (3 * GRU_H) * 4, 0x00020000);
        unsigned ml = (unsigned)(lane & 15);
        asm volatile("" : "+v"(ml)); // formed per use: hoisted, the lane's offsets would be spilled
        const unsigned gl = (unsigned)((((g * RT + rt) * 16 + (int)ml) * T + t) * (3 * GRU_H) + 48 * Jc + 4 * q) * 4u;
#pragma unroll
        for (int k = 0; k < 3; ++k) gpre[k] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_gi, gl + 64u * k, 0, 0));
    };
    // fetch row tile rt of ring rs at byte offset slot_base (of row tile 0 of the batch) into half `b_half` of buffer `buf`
    auto fetch = [&](bool from_h2, unsigned slot_base, int rt, int buf) {
        const unsigned row0 = (unsigned)((g * RT + rt) * GRU_J) * 1024u;
        const unsigned dst = (unsigned)((buf * 2 + (from_h2 ? 1 : 0)) * GRU_J) * 1024u;
        const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(hx1, 0, 4 * n_rt * GRU_J * 1024, 0x00020000);
        const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(hx2, 0, 2 * n_rt * GRU_J * 1024, 0x00020000);
        // (rolled loops: unrolled, the 25 scalar offsets and LDS addresses of every call site were 65 spilled SGPRs)
        if (from_h2) {
#pragma unroll 1
            for (int S = 0; S < GRU_J; ++S)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs2, (__attribute__((address_space(3))) void*)(lds3 + dst + S * 1024), 16, lane16,
                                                         slot_base + row0 + S * 1024, 0, WS_AUX_SC1);
        } else {
#pragma unroll 1
            for (int S = 0; S < GRU_J; ++S)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (__attribute__((address_space(3))) void*)(lds3 + dst + S * 1024), 16, lane16,
                                                         slot_base + row0 + S * 1024, 0, WS_AUX_SC1);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    // wavefront 14: wait (timed first poll) until layer 1 has published need1 steps and layer 2 need2; false on the deadline
    auto poll = [&](unsigned need1, unsigned need2) -> bool {
        const unsigned long long until = __builtin_amdgcn_s_memrealtime() + first_poll_wait;
        while (__builtin_amdgcn_s_memrealtime() < until) __builtin_amdgcn_s_sleep(1);
        const unsigned need = lane < 32 ? need1 : need2;
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (;;) {
            unsigned v = 0xFFFFFFFFu;
            if (need) v = __hip_atomic_load(poll_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__all(v >= need)) return true;
            __builtin_amdgcn_s_sleep(1);
            if (__builtin_amdgcn_s_memrealtime() - t0 > spin_ticks) return false;
        }
    };
    auto give_up = [&]() {
        if (lane == 0) {
            __hip_atomic_store((__attribute__((address_space(1))) unsigned*)err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *s_dead = 1;
        }
    };
    // one chain of one gate: this wavefront's fragments x the row tile's operand in buffer `buf`
    auto chain = [&](int buf) {
        const bool srcB = layer && ws;
        const f32x4* hbp = hb + ((buf * 2 + (srcB ? 1 : 0)) * GRU_J + kp) * 64 + lane;
        f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            if (i < NW - 1 || kp == 0) {
                const f32x4 hv = hbp[i * 128];
                a = MFMA16(w[i].x, hv.x, a);
                a = MFMA16(w[i].y, hv.y, a);
                a = MFMA16(w[i].z, hv.z, a);
                a = MFMA16(w[i].w, hv.w, a);
            }
        }
        const int blk = layer ? (ws ? 0 : 6) + wg * 2 + kp : (tslot * 3 + wg) * 2 + kp;
        unsigned xoff = (unsigned)__builtin_amdgcn_readfirstlane((buf * 12 + blk) * 1024);
        asm volatile("" : "+s"(xoff));
        *reinterpret_cast<f32x4*>(reinterpret_cast<char*>(xch) + xoff + lane16) = a;
    };
    // gate math + publish of (row tile rt, step t) by the tile's helper; products in xch[buf]; layer 1's gi in gpre[slot]
    auto gates = [&](int rt, int t, bool first, int buf) {
        const f32x4* bt = btab + tslot * 24 + q;
        const f32x4* xr = xch + (buf * 12 + (layer ? 0 : tslot * 6)) * 64 + lane; // recurrent products: [gate][chain]
        const f32x4* xi = xch + (buf * 12 + 6) * 64 + lane;                       // layer 2: its input projection
        f32x4 z4, r4, h;
        if (WS_DIAG(variant, 8192)) h = (f32x4){0.f, 0.f, 0.f, 0.f};
        else {
        {
            const f32x4 gi = layer ? (xi[0] + xi[64]) + bt[12] : gpre[0];
            const f32x4 a = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : xr[0] + xr[64];
            const f32x4 b = bt[0];
#pragma unroll
            for (int r = 0; r < 4; ++r) z4[r] = fast_sigmoid(gi[r] + (a[r] + b[r]));
        }
        {
            const f32x4 gi = layer ? (xi[128] + xi[192]) + bt[16] : gpre[1];
            const f32x4 a = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : xr[128] + xr[192];
            const f32x4 b = bt[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) r4[r] = fast_sigmoid(gi[r] + (a[r] + b[r]));
        }
        {
            const f32x4 gi = layer ? (xi[256] + xi[320]) + bt[20] : gpre[2];
            const f32x4 a = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : xr[256] + xr[320];
            const f32x4 b = bt[8];
            const f32x4 hp = hpv[(rt * 2 + tslot) * 64 + lane];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float n = fast_tanh(gi[r] + r4[r] * (a[r] + b[r]));
                h[r] = (1.0f - z4[r]) * n + z4[r] * hp[r];
            }
        }
        }
        hpv[(rt * 2 + tslot) * 64 + lane] = h;
        const int rtg = g * RT + rt;
        if (layer == 0) {
            const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(hx1, 0, 4 * n_rt * GRU_J * 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs1, lane16, (unsigned)((((t & 3) * n_rt + rtg) * GRU_J + J) * 1024), WS_AUX_SC1);
        } else {
            const auto rs2 = __builtin_amdgcn_make_buffer_rsrc(hx2, 0, 2 * n_rt * GRU_J * 1024, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs2, lane16, (unsigned)((((t & 1) * n_rt + rtg) * GRU_J + J) * 1024), WS_AUX_SC1);
        }
    };
    // every tile of the step is stored: the storing wavefront drains, raises its tile's flag, then (layer 2) writes the
    // row-major copy fc2 reads -- not part of the hand-off -- from hpv
    auto publish_step = [&](int t) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_store(my_flag, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (layer) {
            const auto rs_out = __builtin_amdgcn_make_buffer_rsrc(hout2, 0, n_rt * 16 * T * GRU_H * 4, 0x00020000);
            for (int rt = 0; rt < my_rt; ++rt) {
                unsigned l16 = lane16;
                asm volatile("" : "+v"(l16));
                const unsigned out_lane = (unsigned)((g * RT + rt) * 16 * T * GRU_H * 4) + ((l16 >> 4) & 15u) * (unsigned)(T * GRU_H * 4) + (l16 >> 8) * 16u +
                                          (unsigned)(16 * Jc * 4);
                const f32x4 h = hpv[(rt * 2) * 64 + lane];
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, h), rs_out, out_lane, (unsigned)t * (GRU_H * 4), 0);
            }
        }
    };

    // the row tiles of one step: products of row tile rt beside the fetch of rt + 1 and the gate math of rt - 1 (ONE call site
    // of the gate math: inlined three times it cost 12 spilled VGPRs)
    auto step_products = [&](int t, bool useA, unsigned slotA, bool useB, unsigned slotB) {
        const bool first = layer ? !useB : !useA; // h_{t-1} = 0: no recurrent product
        for (int rt = 0; rt <= my_rt; ++rt) {
            const int buf = rt & 1;
            if (rt < my_rt) {
                if (rt + 1 < my_rt && !WS_DIAG(variant, 4096)) {
                    if (wave == 14 && useA) fetch(false, slotA, rt + 1, buf ^ 1);
                    if (wave == 15 && useB) fetch(true, slotB, rt + 1, buf ^ 1);
                }
                if (gate_wave && tile_ok && (layer ? (ws ? useB : useA) : useA) && !WS_DIAG(variant, 16384)) chain(buf);
            }
            if (helper && rt >= 1) {
                gates(rt - 1, t, first, buf ^ 1);
                if (layer == 0 && rt < my_rt) request_gi(rt, t);
            }
            if (rt < my_rt) {
                // The row-tile barrier orders LDS traffic only (partial products, hpv, the operand buffers -- whose LDS-DMA the
                // fetching wavefronts have waited for themselves).  __syncthreads() would also make the helper wait for its
                // just-issued tile store and gi loads -- a memory round trip inside every row tile (measured with that form:
                // 2.18 -> 1.89 ms at 1024 sequences)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        if (helper) publish_step(t);
    };

    if (layer == 0) {
        for (int t = 0; t < T; ++t) {
            const unsigned slotA = (unsigned)(((t - 1) & 3) * n_rt * GRU_J) * 1024u;
            if (helper) request_gi(0, t);
            if (t >= 1) {
                if (wave == 14) {
                    if (poll((unsigned)t, t >= 4 ? (unsigned)(t - 3) : 0u)) fetch(false, slotA, 0, 0);
                    else give_up();
                }
                __syncthreads();
                if (*s_dead) return;
            }
            step_products(t, t >= 1, slotA, false, 0u); // t = 0: gi only
        }
    } else {
        for (int t = 0; t < T; ++t) {
            const unsigned slotA = (unsigned)((t & 3) * n_rt * GRU_J) * 1024u, slotB = (unsigned)(((t - 1) & 1) * n_rt * GRU_J) * 1024u;
            if (wave == 14) {
                if (poll((unsigned)(t + 1), (unsigned)t)) { if (lane == 0) *s_go = t + 1; fetch(false, slotA, 0, 0); }
                else give_up();
            }
            if (wave == 15 && t >= 1) {
                while (*s_go < t + 1 && *s_dead == 0) __builtin_amdgcn_s_sleep(1);
                if (*s_dead == 0) fetch(true, slotB, 0, 0);
            }
            __syncthreads();
            if (*s_dead) return;
            step_products(t, true, slotA, t >= 1, slotB);
        }
    }
}

// geometry of the pipelined launch: G groups of 38 workgroups, RT row tiles per group
bool fvad_gru_ws2_shape(long n_seq_pad, int n_cu, int* RT, int* G)
{
    if (n_seq_pad <= 0 || n_seq_pad % 16) return false;
    const int n_rt = (int)(n_seq_pad / 16);
    const int g_max = n_cu / 38;
    if (g_max < 1) return false;
    const int rt = (n_rt + g_max - 1) / g_max;
    if (rt > 16) return false; // LDS: 125 KB + 2 KB per row tile (gru_ws2m_kernel)
    *RT = rt;
    *G = (n_rt + rt - 1) / rt;
    return true;
}

static bool ws2k_fits(long n_seq_pad, int T, int n_cu, int variant, bool gi1k);
// which kernel a pipelined launch of n_seq_pad sequences of T steps runs: 0 none fits, 1 gru_ws2k (one row tile per group),
// 2 gru_ws2m (2..16 row tiles per group, 32-bit offsets into gi and h2), 3 gru_ws2 (8 wavefronts, up to 4 row tiles)
static int ws2_kernel_for(long n_seq_pad, int T, int n_cu, int variant, bool* gi1k)
{
    int RT = 0, G = 0;
    *gi1k = false;
    if (!fvad_gru_ws2_shape(n_seq_pad, n_cu, &RT, &G)) return 0;
    if (!(variant & 8)) {
        if (RT == 1) {
            // Layer 1's input projection in the kernel pays in groups of 13 + 25 (six row tiles: 81..96 sequences -- BASELINE
            // config 3's 82 chunks: 326 us against 317 + a 28 us GEMM), not in groups of 25 + 25 (up to 80 sequences, every live
            // push: a one-chunk pass 296 us against 270 + 14), where a layer-1 workgroup's product phase is too short to hide
            // it: there the GEMM in front stays.  Same-box A/B: tools/ws2_ab.py.
            const bool groups_13_25 = G * 2 * GRU_J > n_cu || (variant & 16);
            if (!(variant & 1024) && groups_13_25 && ws2k_fits(n_seq_pad, T, n_cu, variant, true)) { *gi1k = true; return 1; }
            if (ws2k_fits(n_seq_pad, T, n_cu, variant, false)) return 1;
        } else if ((long long)n_seq_pad * T * (3 * GRU_H) * 4 < (1ll << 31)) return 2; // gru_ws2m's buffer resources and offsets are plain int arithmetic (like ws2k_fits)
    }
    return RT <= 4 ? 3 : 0;
}

bool fvad_gru_ws2_ok(long n_seq_pad, int T, int n_cu, int variant)
{
    bool k;
    return ws2_kernel_for(n_seq_pad, T, n_cu, variant, &k) != 0;
}

size_t fvad_gru_ws2_exchange_floats(long n_seq_pad) { return (size_t)10 * (size_t)(n_seq_pad / 16) * GRU_J * 256; } // 4 + 2 slots across the chip, 2 + 2 inside an XCD

// Layer 1 of gru_ws2k exchanging inside one XCD: 8 XCCs of 32 CUs (MI355X), one row tile per group, and places for the sets:
// two groups of 13 per XCC (six groups on three XCCs), or one group of 25 (up to five on five).  variant bit 2048 switches it off.
bool fvad_gru_ws2_local_layer1(long n_seq_pad, int T, int n_cu, int variant)
{
    bool k;
    if (n_cu != 256 || (variant & (8 | 64 | 2048)) || ws2_kernel_for(n_seq_pad, T, n_cu, variant, &k) != 1) return false;
    return true;
}

// the 16-wavefront kernel (one row tile per group) computes layer 1's input projection itself: no GEMM launch in front of it.
// It addresses the features and h2 through 32-bit buffer offsets, so a launch whose h2 exceeds 2 GiB keeps the other kernel.
static bool ws2k_fits(long n_seq_pad, int T, int n_cu, int variant, bool gi1k)
{
    int RT = 0, G = 0;
    if (!fvad_gru_ws2_shape(n_seq_pad, n_cu, &RT, &G)) return false;
    // 32-bit buffer offsets into h2 and into the features (GI1K) or gi1 (a few sequences of tens of thousands of steps do not fit)
    const long long per_row = gi1k ? GRU_H : 3 * GRU_H;
    const bool fits32 = (long long)(n_seq_pad / 16) * 16 * T * per_row * 4 < (1ll << 31);
    return RT == 1 && fits32 && !(variant & 8);
}

bool fvad_gru_ws2_gi1_in_kernel(long n_seq_pad, int T, int n_cu, int variant)
{
    bool k;
    return ws2_kernel_for(n_seq_pad, T, n_cu, variant, &k) == 1 && k;
}

const char* fvad_gru_ws2_kernel_name(long n_seq_pad, int T, int n_cu, int variant)
{
    bool k;
    switch (ws2_kernel_for(n_seq_pad, T, n_cu, variant, &k)) {
    case 1: return k ? "gru_ws2k (layers pipelined, both input projections in the kernel)" : "gru_ws2k (layers pipelined)";
    case 2: return "gru_ws2m (layers pipelined, row tiles streamed)";
    case 3: return "gru_ws2 (layers pipelined)";
    default: return "none";
    }
}

int fvad_gru_ws2_wait_class(long n_seq_pad, int T, int n_cu, int variant)
{
    int RT = 0, G = 0;
    bool gi1k = false;
    if (!fvad_gru_ws2_shape(n_seq_pad, n_cu, &RT, &G) || ws2_kernel_for(n_seq_pad, T, n_cu, variant, &gi1k) != 1) return 0;
    const bool one = G * 2 * GRU_J <= n_cu && !(variant & 16);
    return one ? 1 : gi1k ? 3 : 2;
}

unsigned fvad_gru_ws2_builtin_waits(int wait_class, bool local_layer1)
{
    if (local_layer1 && wait_class >= 1 && wait_class <= 3)
        return WS2K_WAIT_L1_LOCAL | ((wait_class == 1 ? WS2K_WAIT_L2_LOCAL_ONE : WS2K_WAIT_L2_LOCAL) << 16);
    switch (wait_class) {
    case 1: return WS2K_WAIT_L1_ONE | (WS2K_WAIT_L2_ONE << 16);
    case 2: return WS2K_WAIT_L1 | (WS2K_WAIT_L2 << 16);
    case 3: return WS2K_WAIT_L1_G | (WS2K_WAIT_L2_G << 16);
    default: return 0;
    }
}

int fvad_launch_gru_ws2(const float* gi1, const float* feat, const float* W1frag, const float* bG1, const float* R1frag, const float* bR1,
                        const float* W2frag, const float* bW2, const float* R2frag, const float* bR2, float* hout2, float* hx,
                        unsigned* flags, unsigned* err, long n_seq_pad, int T, int n_cu, unsigned long long spin_ticks, int variant,
                        unsigned waits, hipStream_t stream, unsigned* lsync)
{
    int RT = 0, G = 0;
    if (!fvad_gru_ws2_shape(n_seq_pad, n_cu, &RT, &G)) return -1;
    const int n_rt = (int)(n_seq_pad / 16);
    float* hx1 = hx;                                    // four slots
    float* hx2 = hx + (size_t)4 * n_rt * GRU_J * 256;   // two slots
    bool gi1k = false;
    const int which = ws2_kernel_for(n_seq_pad, T, n_cu, variant, &gi1k);
    if (which == 0) return -1;
    if (which == 1) { // one row tile per group: K split over 16 wavefronts
        // 158 KB: h1 / h2 operands, partial products, layer 1's 66 KB of stationary input-projection fragments (the kernel's
        // layout constants); more than half of a CU's LDS either way: one workgroup per CU
        const size_t lds_k = (size_t)(2 * GRU_J * 64 + 12 * 64 + 24 * 64 + 2 * 64 + 48 + 176 + 2 * 3 * (kFeatStride / 16) * 64) * 16;
        // up to five row tiles (80 sequences) on 256 CUs: 25 + 25 workgroups per group; six: 13 + 25 (variant 16: always)
        const int nl1 = (G * 2 * GRU_J <= n_cu && !(variant & 16)) ? GRU_J : 13;
        // layer 1 inside one XCD (lsync): every XCC that hosts a layer-1 set must be dealt at least the set's workgroups, and the
        // grid is dealt round-robin over 8 XCCs: at least 8 x 25 workgroups for sets of 25 (the surplus finds no place and leaves)
        const bool local1 = lsync != nullptr && fvad_gru_ws2_local_layer1(n_seq_pad, T, n_cu, variant);
        unsigned n_wg = (unsigned)(G * (nl1 + GRU_J));
        if (local1 && n_wg < 8u * (nl1 == GRU_J ? GRU_J : 26)) n_wg = 8u * (nl1 == GRU_J ? GRU_J : 26); // (two groups of 13 per host XCC)
        const dim3 grid(n_wg);
        float* hx1l = hx + (size_t)6 * n_rt * GRU_J * 256; // behind the four + two cross-XCD slots
        unsigned* ls = local1 ? lsync : nullptr;
        const float* in1 = gi1k ? feat : gi1;
#define WS2K_LAUNCH_(TRACE_, GI1K_, LOCAL_)                                                                                                     \
        {                                                                                                                                       \
            if (hipFuncSetAttribute((const void*)gru_ws2k_kernel<TRACE_, GI1K_, LOCAL_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_k) != hipSuccess) return -2; \
            hipLaunchKernelGGL((gru_ws2k_kernel<TRACE_, GI1K_, LOCAL_>), grid, dim3(1024), lds_k, stream, in1, W1frag, bG1, R1frag, bR1, W2frag, bW2, R2frag, bR2, \
                               hout2, hx1, hx2, flags, flags + 256, err, T, n_rt, spin_ticks, variant, nl1, waits, hx1l, ls);                      \
            return 0;                                                                                                                           \
        }
#define WS2K_LAUNCH(TRACE_, GI1K_) { if (ls) WS2K_LAUNCH_(TRACE_, GI1K_, true) WS2K_LAUNCH_(TRACE_, GI1K_, false) }
#if FVAD_DIAG
        if (variant & 64) { // step trace (tools/ws2_trace.py)
            if (gi1k) WS2K_LAUNCH(true, true)
            WS2K_LAUNCH(true, false)
        }
#endif
        if (gi1k) WS2K_LAUNCH(false, true)
        WS2K_LAUNCH(false, false)
#undef WS2K_LAUNCH
#undef WS2K_LAUNCH_
    }
    if (which == 2) { // 2..16 row tiles per group, streamed through LDS
        const size_t lds_m = (size_t)(4 * GRU_J * 64 + 24 * 64 + RT * 128 + 48 + 1) * 16; // >= 129 KB: one workgroup per CU
        if (hipFuncSetAttribute((const void*)gru_ws2m_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_m) != hipSuccess) return -2;
        hipLaunchKernelGGL(gru_ws2m_kernel, dim3((unsigned)(G * 38)), dim3(1024), lds_m, stream, gi1, R1frag, bR1, W2frag, bW2, R2frag, bR2,
                           hout2, hx1, hx2, flags, flags + 256, err, T, RT, n_rt, spin_ticks, variant, (waits & 0xFFFFu) ? (waits & 0xFFFFu) : WS2M_WAIT);
        return 0;
    }
    // more than half of a CU's 160 KB of LDS: one workgroup per CU (all workgroups of the launch spin on each other)
    const size_t need = (size_t)(4 * GRU_J + RT * 14) * 1024 + 32;
    const size_t lds = need > 84 * 1024 ? need : 84 * 1024;
    if (lds > 160 * 1024) return -1;
    if (hipFuncSetAttribute((const void*)gru_ws2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(gru_ws2_kernel, dim3((unsigned)(G * 38)), dim3(512), lds, stream, gi1, R1frag, bR1, W2frag, bW2, R2frag, bR2,
                       hout2, hx1, hx2, flags, flags + 256, err, T, RT, n_rt, spin_ticks, variant);
    return 0;
}

__global__ void zero_words_kernel(unsigned* p, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}
__global__ void count_word_kernel(unsigned long long* counter, const unsigned* word)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && *word != 0u) *counter += 1ull;
}
void fvad_launch_count_word(unsigned long long* counter, const unsigned* word, hipStream_t stream)
{
    hipLaunchKernelGGL(count_word_kernel, dim3(1), dim3(64), 0, stream, counter, word);
}
void fvad_launch_zero_words(unsigned* p, int n, hipStream_t stream)
{
    hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, n);
}

// Geometry of a launch over n_seq_pad sequences (a multiple of 16) on n_cu compute units: G groups of
// RT row tiles, 25 workgroups per group.  Returns false when the batch is too large for this kernel.
bool fvad_gru_ws_shape(long n_seq_pad, int n_cu, int* RT, int* G)
{
    if (n_seq_pad <= 0 || n_seq_pad % 16) return false;
    const int n_rt = (int)(n_seq_pad / 16);
    const int g_max = n_cu / GRU_J;
    if (g_max < 1) return false;
    const int rt = (n_rt + g_max - 1) / g_max;
    if (rt > 16) return false;
    *RT = rt;
    *G = (n_rt + rt - 1) / rt;
    return true;
}

size_t fvad_gru_ws_exchange_floats(long n_seq_pad) { return (size_t)2 * (size_t)(n_seq_pad / 16) * GRU_J * 256; }

int fvad_launch_gru_ws(const float* gi, const float* R2frag, const float* bR, float* hout, float* hx, unsigned* flags,
                       unsigned* err, long n_seq_pad, int T, int n_cu, int tile_major, unsigned long long spin_ticks,
                       hipStream_t stream)
{
    const int gi_js = tile_major ? 48 : 16, gi_gs = tile_major ? 16 : GRU_H;
    // spin_ticks (100 MHz ticks; context option ws_spin_ticks, default: max(2 ms, 20 x the launch's expected duration),
    // nn_dispatch.cpp ws_spin_deadline): 0 makes every wait that does not
    // succeed at once give up, which is how the tests drive the gru_lat fallback behind this kernel
    int RT = 0, G = 0;
    if (!fvad_gru_ws_shape(n_seq_pad, n_cu, &RT, &G)) return -1;
    const int n_rt = (int)(n_seq_pad / 16);
    // more than half of a CU's 160 KB of LDS: one workgroup per CU, whatever the exchange tile needs
    const size_t need = (size_t)(2 * GRU_J + RT * 3) * 1024 + 16;
    const size_t lds = need > 84 * 1024 ? need : 84 * 1024;
#define WS_CASE(OWN_)                                                                                               \
    {                                                                                                               \
        if (hipFuncSetAttribute((const void*)gru_ws_kernel<OWN_>, hipFuncAttributeMaxDynamicSharedMemorySize,       \
                                (int)lds) != hipSuccess) return -2;                                                 \
        hipLaunchKernelGGL((gru_ws_kernel<OWN_>), dim3((unsigned)(G * GRU_J)), dim3(256), lds, stream, gi, R2frag,  \
                           bR, hout, hx, flags, err, T, RT, n_rt, gi_js, gi_gs, spin_ticks);                        \
        return 0;                                                                                                   \
    }
    if (RT <= 4) WS_CASE(1)
    if (RT <= 8) WS_CASE(2)
    WS_CASE(4)
#undef WS_CASE
}
