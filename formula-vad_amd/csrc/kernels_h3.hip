// kernels_h3.hip -- NSNet2's matrix products on the f16 matrix cores with f32-class accuracy ("f16x3").
//
// The f32 MFMA (v_mfma_f32_16x16x4_f32) runs at 1/16 of the f16 rate, and the whole network sits on it at
// 0.86 of its peak (kernels_nn.hip).  Here every f32 operand is split into two f16 pieces of a power-of-two
// scaled value,
//     x * 2^Px = xh + xl        xh = f16(x 2^Px),  xl = f16(x 2^Px - xh)      (both subtractions exact in f32)
// which keeps 22 significand bits (f32: 24) as long as |x 2^Px| stays in f16's normal range; the scales come from
// rigorous bounds fixed at model-load time (engine.cpp, h3_scales): |h| < 1 for the GRU states, l1 row norms for
// the dense layers, log10(FLT_MAX^2) for the features.  A product then needs three f16 MFMAs with f32 accumulation
//     W x ~ (Wh xh + Wh xl + Wl xh) 2^-(Pw+Px)
// (f16 x f16 products are exact in f32; the dropped Wl xl term is 2^-22 relative), i.e. 3 x 16 cycles per
// 16x16x32 block against 8 x 32 cycles of the f32 instruction: 5.3 x the f32 matrix rate.  Measured against
// float64 the result is as close as the f32 kernels' (tests/test_gpu.py::test_f16x3_*): the error is dominated
// by the f32 accumulation either way.
//
// Operand convention ("row panel", as in kernels_nn.hip): a wavefront owns 16 activation rows per row tile;
// weights are the A operand, activations the B operand of v_mfma_f32_16x16x32_f16, so lane (m = l & 15,
// q = l >> 4) ends up with output units 16 T + 4 q + {0..3} of row m.  The reduction index inside a 32-deep
// K-step S is permuted so that the same lane's B fragment is exactly two such float4s:
//     slot (q, j):   k = 32 S + 4 q + j            (j < 4,   from unit tile 2 S)
//                    k = 32 S + 16 + 4 q + (j - 4) (j >= 4,  from unit tile 2 S + 1)
// (the MFMA only sums over k, so any bijection works as long as the weight fragments use the same one:
// tables_weights.cpp pack_panel_h3).  Weight fragment block (T, S, piece) = 64 lanes x 8 halves = 1 KB,
// [T-major inside a K-step][hi, lo].

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.h"
#include "nn_device.h"

typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)

namespace {

template <int T, int N> struct SFor {
    template <class F> static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, T>{});
        SFor<T + 1, N>::run(f);
    }
};
template <int N> struct SFor<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&&) {}
};

template <int OFF> __device__ __forceinline__ void lds_read_b128(f32x4& dst, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait2(f32x4& a, f32x4& b)
{
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N));
}

__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// two float4s (k-slots j = 0..3 and 4..7 of this lane) -> the hi and lo f16 fragments of the scaled values
__device__ __forceinline__ void split8(const f32x4& a, const f32x4& b, float sx, h16x8& hi, h16x8& lo)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float va = a[j] * sx, vb = b[j] * sx;
        const _Float16 ha = (_Float16)va, hb = (_Float16)vb;
        hi[j] = ha;
        hi[4 + j] = hb;
        lo[j] = (_Float16)(va - (float)ha);
        lo[4 + j] = (_Float16)(vb - (float)hb);
    }
}

} // namespace

// ------------------------------------------------------------------ panel GEMM, f16x3 (persistent)
// Same skeleton as panel_gemm3_kernel: one workgroup per CU walks (row panel of WAVES*RT row tiles, column
// block of NT tiles) items; weight fragment blocks stream through two LDS slabs by LDS-DMA, SP K-steps per
// phase; fragment reads are inline ds_read_b128 with counted lgkmcnt waits, a ring of D tiles ahead
// (D divides NT so that ring slots are compile-time).  Activations are loaded as f32 two K-steps ahead and
// split at the end of a step, next to the other wavefront's MFMAs.
//
// Timing-only variants and a device-clock timeline of the first version (f32 row-major activations, split
// in the loop) showed where its time went: a K-step's 90 MFMAs issue in ~1 450 cycles, but each wavefront then
// spent ~800 cycles splitting the next step's operands and ~500 waiting for them (16 rows x 64 bytes per
// load instruction, queued behind the weight DMA), ~700 issuing the DMA, and the partner wavefront of the SIMD
// was usually in the same state.  So the f16x3 path keeps its intermediates in TILED layouts:
//   * rows are regrouped into row tiles of 16 SEQUENCES at one time step,
//         row tile R = group * T + t   holds rows (sequence 16 group + m, time t), m = 0..15,
//     which is also what the recurrence needs (16 sequences at the same t);
//   * "TL" (f32, gi only): [R][unit tile U][64 lanes][4 floats], lane = m + 16 q, float r = unit 16 U + 4 q + r
//     -- one MFMA result tile per 1 KB block;
//   * "TS" (split, every matrix-product INPUT: h1, h2, f2, f3): [R][K-step S][hi, lo][64 lanes][8 halves] -- the
//     two f16 fragments of a K-step exactly as the MFMA takes them, already scaled by the consumer's 2^Px.  The
//     producer (GEMM epilogue / recurrence) does the split once per value; the consumer's K loop is loads and
//     MFMAs only, and every load or store wave-instruction moves 1 KB of contiguous memory.
// Row-major buffers exist only at the ends: the features the STFT kernel writes (IN_TS = false: the 16 rows of
// a tile are gathered and split in the loop, K = 161 only) and the gains the inverse STFT kernel reads
// (OUT = 0: scattered through the lane transpose of panel_gemm3_kernel).
//   S_steps: 32-deep K-steps; k_tiles = ceil(K / 16) (the upper half of the last K-step is zero when it is odd).
//   a_ld: floats per row (row-major) or K-steps per row tile (TS).  c_ld: floats per row (OUT 0), unit tiles per
//   row tile (OUT 1, TL) or K-steps per row tile (OUT 2, TS).  seq_T: rows per sequence of the row-major side.
//   map_T / map_skip: output row tile Ro reads input row tile
//   (Ro / (map_T - map_skip)) * map_T + map_skip + Ro % (map_T - map_skip)  (fc2 skips the 4 warm-up steps).
//   bias_scale = 2^(Pw+Px), out_scale = 2^-(Pw+Px); out_sx: the next layer's 2^Px (OUT 2).
template <int NT, int RT, int ACT, int SP, int D, int WAVES, bool IN_TS, int OUT>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void panel_gemm_h3_kernel(
    const float* __restrict__ A, int a_ld, const float* __restrict__ Wfrag,
    const float* __restrict__ bias, float* __restrict__ C, int c_ld, int seq_T, int S_steps, int k_tiles, int n_blocks,
    int n_valid_tiles, int map_T, int map_skip, unsigned n_items, float sx, float bias_scale, float out_scale,
    float out_sx)
{
    static_assert(NT % D == 0, "ring slots must be compile-time");
    static_assert(OUT != 2 || NT % 2 == 0, "the split output pairs unit tiles");
    static_assert(NT * SP * 2048 <= 65536, "ds_read offset field");
    __shared__ __attribute__((aligned(16))) float slab[2][NT * SP * 512];
    __shared__ __attribute__((aligned(16))) float sbias[2][NT * 16];
    typedef __attribute__((address_space(3))) float lds_float;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    const int P = (S_steps + SP - 1) / SP;

    auto a_pointer = [&](unsigned item, int rt) -> const float* {
        const unsigned panel = item / (unsigned)n_blocks;
        unsigned R = (panel * WAVES + wave) * RT + rt;
        if (map_T > 0) {
            const unsigned per = (unsigned)(map_T - map_skip);
            const unsigned g = R / per;
            R = g * (unsigned)map_T + (unsigned)map_skip + (R - g * per);
        }
        if (IN_TS) return A + (size_t)R * (size_t)a_ld * 512 + lane * 4;
        const unsigned g = R / (unsigned)seq_T, t = R - g * (unsigned)seq_T;
        return A + ((size_t)(g * 16 + m) * seq_T + t) * (size_t)a_ld + 4 * q;
    };
    auto issue = [&](unsigned item, int p, float* dst) {
        const int nblk = (int)(item % (unsigned)n_blocks);
        const int s0 = p * SP;
        const int cnt = (S_steps - s0 < SP) ? (S_steps - s0) : SP;
        const float* src = Wfrag + ((size_t)nblk * S_steps + s0) * (NT * 512);
        const int nb = NT * cnt * 2; // 1 KB blocks
        constexpr int PER = (NT * SP * 2 + WAVES - 1) / WAVES;
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            int b = wave + i * WAVES;
            b = b < nb ? b : nb - 1;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + b * 256 + lane * 4),
                                             (__attribute__((address_space(3))) void*)(dst + b * 256), 16, 0, 0);
        }
    };
    auto stage_bias = [&](unsigned item, float* dst) {
        const int nblk = (int)(item % (unsigned)n_blocks);
        if (tid < NT * 4) {
            f32x4 b4 = reinterpret_cast<const f32x4*>(bias + nblk * (NT * 16))[tid];
            b4 *= bias_scale;
            reinterpret_cast<f32x4*>(dst)[tid] = b4;
        }
    };

    unsigned item = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    if (item >= n_items) return;
#ifdef H3_STAGGER
    // Workgroups do identical work, so left alone they all reach their epilogues together and the chip's
    // stores (and the activation reads) come in bursts; starting them up to one item apart spreads the traffic.
    for (int i = 0, n = (int)((blockIdx.x / 8u) % 16u) * S_steps * H3_STAGGER / 16; i < n; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    const unsigned slab_addr[2] = {(unsigned)(uintptr_t)(lds_float*)slab[0] + (unsigned)lane * 16u,
                                   (unsigned)(uintptr_t)(lds_float*)slab[1] + (unsigned)lane * 16u};
    typedef const __attribute__((address_space(1))) f32x4* gptr4;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one K-step of this lane's operand: TS: the two fragments as stored; row-major: two float4s to be split
    auto load_step = [&](const float* base, int s, f32x4& a4, f32x4& b4) {
        if (IN_TS) {
            a4 = *(gptr4)(base + 512 * s);
            b4 = *(gptr4)(base + 512 * s + 256);
        } else {
            a4 = *(gptr4)(base + 32 * s);
            b4 = (2 * s + 1 < k_tiles) ? *(gptr4)(base + 32 * s + 16) : zero4;
        }
    };
    auto to_frags = [&](const f32x4& a4, const f32x4& b4, h16x8& hi, h16x8& lo) {
        if (IN_TS) { hi = __builtin_bit_cast(h16x8, a4); lo = __builtin_bit_cast(h16x8, b4); }
        else split8(a4, b4, sx, hi, lo);
    };

    const float* a_ptr[RT];
    h16x8 xh[RT], xl[RT];   // this K-step's activation fragments
    f32x4 ra[RT], rb[RT];   // the K-step after it, as loaded
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        a_ptr[rt] = a_pointer(item, rt);
        f32x4 a0, b0;
        load_step(a_ptr[rt], 0, a0, b0);
        to_frags(a0, b0, xh[rt], xl[rt]);
        load_step(a_ptr[rt], 1, ra[rt], rb[rt]);
    }
    issue(item, 0, slab[0]);
    stage_bias(item, sbias[0]);
    __syncthreads();
    int buf = 0, bbuf = 0;

    for (; item < n_items; item += gridDim.x) {
        const unsigned next_item = item + gridDim.x;
        const bool has_next = next_item < n_items;
        f32x4 acc[RT][NT];
        {
            const f32x4* bl = reinterpret_cast<const f32x4*>(sbias[bbuf]) + q;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const f32x4 b4 = bl[4 * t];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) acc[rt][t] = b4;
            }
        }
        const float* a_nextitem[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a_nextitem[rt] = has_next ? a_pointer(next_item, rt) : a_ptr[rt];

        for (int p = 0; p < P; ++p) {
            const int s0 = p * SP;
            const int cnt = (S_steps - s0 < SP) ? (S_steps - s0) : SP;
            if (p + 1 < P) issue(item, p + 1, slab[buf ^ 1]);
            else if (has_next) { issue(next_item, 0, slab[buf ^ 1]); stage_bias(next_item, sbias[bbuf ^ 1]); }

            unsigned rd = slab_addr[buf];
            f32x4 wh[D], wl[D];
            SFor<0, D>::run([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                lds_read_b128<t * 2048>(wh[t], rd);
                lds_read_b128<t * 2048 + 1024>(wl[t], rd);
            });
            for (int s = 0; s < cnt; ++s) {
                const int sg = s0 + s;
                // in the last step of a phase the tiles past the ring's reach are in the other buffer, behind
                // the barrier: read this step's tile again so that the lgkmcnt arithmetic stays uniform
                const unsigned rdn = (s + 1 < cnt) ? rd : rd - NT * 2048; // for tiles of the NEXT step only
                SFor<0, NT>::run([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int k = t % D;
                    lds_wait2<2 * (D - 1)>(wh[k], wl[k]);
                    const h16x8 ah = __builtin_bit_cast(h16x8, wh[k]);
                    const h16x8 al = __builtin_bit_cast(h16x8, wl[k]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_H(al, xh[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_H(ah, xl[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_H(ah, xh[rt], acc[rt][t]);
                    const unsigned ra_ = (t + D < NT) ? rd : rdn;
                    lds_read_b128<(t + D) * 2048>(wh[k], ra_);
                    lds_read_b128<(t + D) * 2048 + 1024>(wl[k], ra_);
                });
                // next K-step's fragments from the f32 values requested one step ago, then the request for the
                // step after it (the next item's first steps near the end: in flight across the epilogue)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    to_frags(ra[rt], rb[rt], xh[rt], xl[rt]);
                    const int s2 = sg + 2;
                    if (s2 < S_steps) load_step(a_ptr[rt], s2, ra[rt], rb[rt]);
                    else load_step(a_nextitem[rt], s2 - S_steps, ra[rt], rb[rt]);
                }
                rd += NT * 2048;
            }
            // the last step's look-ahead reads are still in flight and the compiler does not know (it considers the
            // ring registers dead from here on): wait for them while they are still allocated
            SFor<0, D>::run([&](auto tc) {
                constexpr int k = decltype(tc)::value;
                lds_wait2<0>(wh[k], wl[k]);
            });
            __syncthreads();
            buf ^= 1;
        }

        const int nblk = (int)(item % (unsigned)n_blocks);
        const unsigned panel = item / (unsigned)n_blocks;
        const int valid_t = n_valid_tiles - nblk * NT;
        auto activate = [&](f32x4 v) {
            v *= out_scale;
            if (ACT == FVAD_ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (ACT == FVAD_ACT_SIGMOID) {
                v.x = act_sigmoid(v.x); v.y = act_sigmoid(v.y); v.z = act_sigmoid(v.z); v.w = act_sigmoid(v.w);
            }
            return v;
        };
        float* c_ptr[RT];
        const int bp_addr = ((lane >> 2) + 16 * (lane & 3)) * 4; // row-major output: lane 4 m + q takes row m, columns 4 q ..
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned R = (panel * WAVES + wave) * RT + rt;
            if (OUT == 1) {
                c_ptr[rt] = C + ((size_t)R * (size_t)c_ld + (size_t)nblk * NT) * 256 + lane * 4;
            } else if (OUT == 2) {
                c_ptr[rt] = C + ((size_t)R * (size_t)c_ld + (size_t)(nblk * NT / 2)) * 512 + lane * 4;
            } else {
                const unsigned g = R / (unsigned)seq_T, t = R - g * (unsigned)seq_T;
                c_ptr[rt] = C + ((size_t)(g * 16 + (unsigned)(lane >> 2)) * seq_T + t) * (size_t)c_ld + nblk * (NT * 16) + 4 * (lane & 3);
            }
        }
        auto emit = [&](int t, int rt) {
            f32x4 v = activate(acc[rt][t]);
            if (OUT == 1) {
                *reinterpret_cast<f32x4*>(c_ptr[rt] + 256 * t) = v;
            } else {
                f32x4 o;
                o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[0])));
                o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[1])));
                o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[2])));
                o.w = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[3])));
                *reinterpret_cast<f32x4*>(c_ptr[rt] + 16 * t) = o;
            }
        };
        // split output: unit tiles 2 u and 2 u + 1 of this lane are the two halves of K-step u's fragments
        auto emit_pair = [&](int u, int rt) {
            const f32x4 v0 = activate(acc[rt][2 * u]), v1 = activate(acc[rt][2 * u + 1]);
            h16x8 hi, lo;
            split8(v0, v1, out_sx, hi, lo);
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 512 * u) = __builtin_bit_cast(f32x4, hi);
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 512 * u + 256) = __builtin_bit_cast(f32x4, lo);
        };
        if (OUT == 2) {
#pragma unroll
            for (int u = 0; u < NT / 2; ++u) {
                if (2 * u < valid_t) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) emit_pair(u, rt);
                }
            }
        } else if (valid_t >= NT) {
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) emit(t, rt);
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                if (t < valid_t) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) emit(t, rt);
                }
            }
        }
        bbuf ^= 1;
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a_ptr[rt] = a_nextitem[rt];
    }
}

// row_tiles (output row tiles of 16 rows) must be a multiple of 16; grid = one persistent workgroup per CU.
// in_ts: A is in the split tiled layout (a_ld = K-steps per row tile) or row-major f32 [sequence][seq_T][a_ld];
// out: 0 row-major f32 [sequence][seq_T][c_ld], 1 TL f32 (c_ld unit tiles per row tile), 2 TS (c_ld K-steps per
// row tile, scaled by out_sx).  K = true reduction length.  Returns -1 when there is no instance.
int fvad_launch_panel_gemm_h3(const float* A, int in_ts, int a_ld, const float* Wfrag, const float* bias, float* C,
                              int out, int c_ld, int seq_T, long row_tiles, int nt, int n_blocks, int K, int act,
                              int n_valid_tiles, int map_T, int map_skip, float sx, float sw, float out_sx, int n_wg,
                              hipStream_t stream)
{
    if (row_tiles % 16) return -1;
    const int k_tiles = (K + 15) / 16;
    const int S_steps = (k_tiles + 1) / 2;
    if (S_steps < 3) return -1;
    unsigned n_items = (unsigned)((row_tiles / 16) * n_blocks);
    unsigned grid = n_items < (unsigned)n_wg ? n_items : (unsigned)n_wg;
    const float bias_scale = sx * sw, out_scale = 1.0f / (sx * sw);
    // (A three-row-tile form of the fc2 / fc3 instance was 5 % faster, but it sits one register over the 256-register cap:
    // one spilled VGPR.  A spill is not slow here, it is unsafe -- the inline ds_reads and the LDS-DMA land in registers
    // the compiler believes it may reuse -- so the form is gone.)
#define CASEH(NT_, ACT_, SP_, D_, IN_, OUT_)                                                                  \
    if (nt == NT_ && act == ACT_ && (in_ts != 0) == IN_ && out == OUT_) {                                     \
        hipLaunchKernelGGL((panel_gemm_h3_kernel<NT_, 2, ACT_, SP_, D_, 8, IN_, OUT_>), dim3(grid), dim3(512), 0, \
                           stream, A, a_ld, Wfrag, bias, C, c_ld, seq_T, S_steps, k_tiles, n_blocks,          \
                           n_valid_tiles, map_T, map_skip, n_items, sx, bias_scale, out_scale, out_sx);       \
        return 0;                                                                                             \
    }
    CASEH(15, FVAD_ACT_NONE, 2, 3, false, 1)   // features (row-major) -> gi (TL)
    CASEH(15, FVAD_ACT_NONE, 2, 5, true, 1)    // h1 (TS) -> gi (TL)
    CASEH(10, FVAD_ACT_RELU, 3, 5, true, 2)    // fc2, fc3 (TS -> TS)
    CASEH(12, FVAD_ACT_SIGMOID, 2, 4, true, 0) // fc4 (TS) -> gains (row-major)
#undef CASEH
    return -1;
}

// ------------------------------------------------------------------ GRU recurrence, f16x3
// gru_rec3_kernel's structure (kernels_nn.hip) on the f16 matrix cores: a wavefront owns 16 sequences and keeps
// h_{t-1} in registers, now as the 13 x (hi, lo) f16 fragments of h 2^14 (104 VGPRs; the upper half of K-step 12
// is zero); a unit tile's recurrent weights -- [13 S][3 gates][hi, lo] fragment blocks, 78 KB, pack_gru_r_h3 --
// stream through two LDS slabs by LDS-DMA while the previous tile computes: 9 MFMAs per K-step
//     a_g += Wg_hi hl + Wg_hi hh + Wg_lo hh        g = z, r, n
// i.e. 117 MFMAs of 16 cycles per tile against 300 of 32.  The gates are gru_rec3's.  gi is in the tiled f32
// layout of panel_gemm_h3_kernel; h is written, and read back at the top of the next step, as split fragments.
constexpr int H3_S = 13;                       // 32-deep K-steps covering H = 400 (416 slots)
constexpr int H3_SLAB = H3_S * 3 * 512;        // floats per unit tile: 79872 bytes

template <int WAVES>
__device__ __forceinline__ void h3_issue_slab(const float* __restrict__ src, float* lds_dst, int wave, unsigned lane16)
{
    asm volatile("" : "+v"(lane16));
#pragma unroll
    for (int b = wave; b < H3_S * 6; b += WAVES) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((const __attribute__((address_space(1))) char*)src + b * 1024 + lane16),
            (__attribute__((address_space(3))) void*)(lds_dst + b * 256), 16, 0, 0);
    }
}

template <int N> __device__ __forceinline__ void lds_wait1(f32x4& a)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N));
}

template <int WAVES, int D>
__global__ __launch_bounds__(WAVES * 64) void gru_rec_h3_kernel(const float* __restrict__ gi,
                                                                const float* __restrict__ Rfrag,
                                                                const float* __restrict__ bR,
                                                                float* hsplit, int T,
                                                                float sx, float bias_scale, float out_scale)
{
    __shared__ __attribute__((aligned(16))) float slab[2][H3_SLAB];
    // the n gate's recurrent bias: read from LDS in the gate math, after the tile's last MFMA (a global load at the top of
    // every unit tile costs an address register in the hot loop: 4 VGPR spills at the 168-register cap of three
    // wavefronts per SIMD; and an LDS read up there would sit in front of the counted waits of the fragment ring)
    __shared__ __attribute__((aligned(16))) float sbn[GRU_H];
    typedef __attribute__((address_space(3))) float lds_float;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    for (int i = tid; i < GRU_H; i += WAVES * 64) sbn[i] = bR[2 * GRU_H + i];
    (void)bias_scale;
    // gi and hout are in the tiled layout (see panel_gemm_h3_kernel): this wavefront's 16 sequences are group
    // `grp`; row tile grp * T + t holds their time step t as [unit tile][64 lanes][4 floats] blocks, so a gate's
    // operand tile, h_{t-1}'s operand halves and the h_t store are 1 KB of contiguous memory each.  gi's unit
    // tiles are in TILE-major gate order: 3 J + {z, r, n}.
    typedef const __attribute__((address_space(1))) char* gbytes;
    const size_t grp = (size_t)(blockIdx.x * WAVES + wave);
    gbytes gi_w = (gbytes)(gi + grp * T * (3 * GRU_J * 256));
    // h lives only in the split tiled layout (TS): the f16 pieces of h 2^14 are what this kernel's next step and
    // the next matrix product take as operands, and the z h_{t-1} term reads h_{t-1} back as hi + lo -- 22
    // significand bits instead of 24, which changes the gains by less than the f32 accumulation does (emulated:
    // rms error against float64 8.2e-8 -> 8.3e-8, maximum unchanged)
    __attribute__((address_space(1))) char* hs_w = (__attribute__((address_space(1))) char*)(hsplit + grp * T * (H3_S * 512));
    gbytes bR_b = (gbytes)bR;
    const unsigned gi_off = (unsigned)lane * 16u;
    const unsigned h_off = (unsigned)lane * 16u;
    const unsigned b_off = 16u * q;
    const float inv_sx = 1.0f / sx;
    (void)m;
    auto ld4 = [](gbytes base, unsigned off) {
        asm volatile("" : "+v"(off));
        return *(const __attribute__((address_space(1))) f32x4*)(base + off);
    };
    const unsigned slab_addr[2] = {(unsigned)(uintptr_t)(lds_float*)slab[0] + (unsigned)lane * 16u,
                                   (unsigned)(uintptr_t)(lds_float*)slab[1] + (unsigned)lane * 16u};
    // unit tile J of h_t as f16 pieces: half (J & 1) of K-step J / 2's fragments (8 bytes of hi, 8 of lo per
    // lane); the last tile also writes the zero upper half of K-step 12
    typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
    auto store_split = [&](__attribute__((address_space(1))) char* hs_t, int J, const f32x4& h) {
        h16x4 hi, lo;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = h[r] * sx;
            const _Float16 a = (_Float16)v;
            hi[r] = a;
            lo[r] = (_Float16)(v - (float)a);
        }
        unsigned o = (unsigned)lane * 16u;
        asm volatile("" : "+v"(o));
        __attribute__((address_space(1))) char* dst = hs_t + (J >> 1) * 2048 + (J & 1) * 8 + o;
        if (J == GRU_J - 1) {
            const h16x4 z4 = (h16x4){0, 0, 0, 0};
            h16x8 hi8 = __builtin_shufflevector(hi, z4, 0, 1, 2, 3, 4, 5, 6, 7);
            h16x8 lo8 = __builtin_shufflevector(lo, z4, 0, 1, 2, 3, 4, 5, 6, 7);
            *(__attribute__((address_space(1))) h16x8*)dst = hi8;
            *(__attribute__((address_space(1))) h16x8*)(dst + 1024) = lo8;
        } else {
            *(__attribute__((address_space(1))) h16x4*)dst = hi;
            *(__attribute__((address_space(1))) h16x4*)(dst + 1024) = lo;
        }
    };

    h3_issue_slab<WAVES>(Rfrag, slab[0], wave, (unsigned)lane * 16u);

    // ---- t = 0: h_{-1} = 0, so R h + Rb = Rb
    for (int J = 0; J < GRU_J; ++J) {
        const f32x4 giz = ld4(gi_w + 3072 * J, gi_off);
        const f32x4 gir = ld4(gi_w + 3072 * J + 1024, gi_off);
        const f32x4 gin = ld4(gi_w + 3072 * J + 2048, gi_off);
        const f32x4 bn = ld4(bR_b + 64 * J + 8 * GRU_H, b_off);
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = fast_sigmoid(giz[r]);
            const float rr = fast_sigmoid(gir[r]);
            const float n = fast_tanh(gin[r] + rr * bn[r]);
            h[r] = (1.0f - z) * n + z * 0.0f;
        }
        store_split(hs_w, J, h);
    }
    __syncthreads();
    int buf = 0;

    for (int t = 1; t < T; ++t) {
        gbytes gi_t = gi_w + (size_t)t * (3 * GRU_J * 1024);

        gbytes hs_prev = (gbytes)hs_w + (size_t)(t - 1) * (H3_S * 2048);
        __attribute__((address_space(1))) char* hs_out = hs_w + (size_t)t * (H3_S * 2048);
        h16x8 hh[H3_S], hl[H3_S];
#pragma unroll
        for (int S = 0; S < H3_S; ++S) {
            hh[S] = __builtin_bit_cast(h16x8, ld4(hs_prev + 2048 * S, h_off));
            hl[S] = __builtin_bit_cast(h16x8, ld4(hs_prev + 2048 * S + 1024, h_off));
        }
#pragma unroll
        for (int S = 0; S < H3_S; ++S) asm volatile("" : "+v"(hh[S]), "+v"(hl[S]));

        for (int J = 0; J < GRU_J; ++J) {
            const int nJ = (J + 1 == GRU_J) ? 0 : J + 1;
            h3_issue_slab<WAVES>(Rfrag + (size_t)nJ * H3_SLAB, slab[buf ^ 1], wave, (unsigned)lane * 16u);

            // ds_read's offset field is 16 bits and a slab is 78 KB: one base for K-steps 0..7, one for 8..12
            const unsigned rd0 = slab_addr[buf], rd1 = rd0 + 8 * 6144;
            // ring of D K-steps; slot S % D holds z_hi, r_hi, n_hi, z_lo, r_lo, n_lo of K-step S
            f32x4 w[D][6];
            auto read_step = [&](auto sc, auto kc) {
                constexpr int S = decltype(sc)::value;
                constexpr int k = decltype(kc)::value;
                constexpr int base = (S < 8 ? S : S - 8) * 6144;
                const unsigned rd = S < 8 ? rd0 : rd1;
                lds_read_b128<base + 0 * 2048>(w[k][0], rd);
                lds_read_b128<base + 1 * 2048>(w[k][1], rd);
                lds_read_b128<base + 2 * 2048>(w[k][2], rd);
                lds_read_b128<base + 0 * 2048 + 1024>(w[k][3], rd);
                lds_read_b128<base + 1 * 2048 + 1024>(w[k][4], rd);
                lds_read_b128<base + 2 * 2048 + 1024>(w[k][5], rd);
            };
            SFor<0, D>::run([&](auto sc) { read_step(sc, std::integral_constant<int, decltype(sc)::value % D>{}); });

            f32x4 az = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 ar = az, an = az;
            auto k_step = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                constexpr int k = S % D;
                constexpr int ahead = (H3_S - 1 - S < D - 1) ? H3_S - 1 - S : D - 1;
                constexpr int y = 6 * ahead; // reads of younger K-steps in flight
                const h16x8 xh = hh[S], xl = hl[S];
                lds_wait1<(y + 5 < 15 ? y + 5 : 15)>(w[k][0]);
                az = MFMA_H(__builtin_bit_cast(h16x8, w[k][0]), xl, az);
                lds_wait1<(y + 4 < 15 ? y + 4 : 15)>(w[k][1]);
                ar = MFMA_H(__builtin_bit_cast(h16x8, w[k][1]), xl, ar);
                lds_wait1<(y + 3 < 15 ? y + 3 : 15)>(w[k][2]);
                an = MFMA_H(__builtin_bit_cast(h16x8, w[k][2]), xl, an);
                az = MFMA_H(__builtin_bit_cast(h16x8, w[k][0]), xh, az);
                ar = MFMA_H(__builtin_bit_cast(h16x8, w[k][1]), xh, ar);
                an = MFMA_H(__builtin_bit_cast(h16x8, w[k][2]), xh, an);
                lds_wait1<(y + 2 < 15 ? y + 2 : 15)>(w[k][3]);
                az = MFMA_H(__builtin_bit_cast(h16x8, w[k][3]), xh, az);
                lds_wait1<(y + 1 < 15 ? y + 1 : 15)>(w[k][4]);
                ar = MFMA_H(__builtin_bit_cast(h16x8, w[k][4]), xh, ar);
                lds_wait1<(y < 15 ? y : 15)>(w[k][5]);
                an = MFMA_H(__builtin_bit_cast(h16x8, w[k][5]), xh, an);
                if constexpr (S + D < H3_S) read_step(std::integral_constant<int, S + D>{}, std::integral_constant<int, k>{});
            };
            constexpr int LOAD_AT = 9;
            SFor<0, LOAD_AT>::run(k_step);
            const f32x4 giz = ld4(gi_t + 3072 * J, gi_off);
            const f32x4 gir = ld4(gi_t + 3072 * J + 1024, gi_off);
            const f32x4 gin = ld4(gi_t + 3072 * J + 2048, gi_off);
            // h_{t-1} of this unit tile: half (J & 1) of K-step J / 2's fragments
            typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
            h16x4 hp_hi, hp_lo;
            {
                unsigned o = h_off;
                asm volatile("" : "+v"(o));
                gbytes src = hs_prev + (J >> 1) * 2048 + (J & 1) * 8 + o;
                hp_hi = *(const __attribute__((address_space(1))) h16x4*)src;
                hp_lo = *(const __attribute__((address_space(1))) h16x4*)(src + 1024);
            }
            SFor<LOAD_AT, H3_S>::run(k_step);
            const f32x4 bn = *reinterpret_cast<const f32x4*>(sbn + 16 * J + 4 * q);
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fast_sigmoid(giz[r] + az[r] * out_scale);
                const float rr = fast_sigmoid(gir[r] + ar[r] * out_scale);
                const float n = fast_tanh(gin[r] + rr * (an[r] * out_scale + bn[r]));
                const float hp = ((float)hp_hi[r] + (float)hp_lo[r]) * inv_sx;
                h[r] = (1.0f - z) * n + z * hp;
            }
            store_split(hs_out, J, h);
            // the slab DMA is older than this tile's two stores
            asm volatile("s_waitcnt vmcnt(2)\n\ts_barrier" ::: "memory");
            buf ^= 1;
        }
    }
}

// Rfrag: pack_gru_r_h3 layout; sx / sw: the scales of h (2^14) and of R; gi: TL; hsplit: TS with 13 K-steps
int fvad_launch_gru_rec_h3(const float* gi, const float* Rfrag, const float* bR, float* hsplit,
                           long n_seq_pad, int T, int waves, float sx, float sw, hipStream_t stream)
{
    if (waves <= 0 || n_seq_pad % (16 * waves)) return -1;
    const unsigned grid = (unsigned)(n_seq_pad / (16 * waves));
    const float out_scale = 1.0f / (sx * sw);
    if (waves == 8) {
        hipLaunchKernelGGL((gru_rec_h3_kernel<8, 2>), dim3(grid), dim3(512), 0, stream, gi, Rfrag, bR, hsplit, T, sx, sx * sw, out_scale);
        return 0;
    }
    if (waves == 12) {
        hipLaunchKernelGGL((gru_rec_h3_kernel<12, 1>), dim3(grid), dim3(768), 0, stream, gi, Rfrag, bR, hsplit, T, sx, sx * sw, out_scale);
        return 0;
    }
    return -1;
}
