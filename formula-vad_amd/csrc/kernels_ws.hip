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

constexpr int WS_AUX_SC1 = 16;                            // buffer cache-policy bit: sc1 (system-coherent level 1)

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
        // ---- wait until every workgroup of the group has published h_{t-1} (flag >= t)
        if (wave == 3) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
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
    // spin_ticks (100 MHz ticks; context option ws_spin_ticks, default 0.25 s): 0 makes every wait that does not
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
