// kernels_b3.hip -- NSNet2's dense matrix products on the bf16 matrix cores WITHOUT giving up operand bits ("bf16x3").
//
// kernels_h3.hip splits an f32 operand into two f16 pieces: 22 significand bits, narrower than the reference's f32
// (src/NSNet2.zig:220 runs f32 GEMMs), and it needs power-of-two scales fixed from bounds so that nothing leaves
// f16's range.  Here every f32 operand becomes THREE bf16 pieces,
//     x = xh + xm + xl      xh = bf16(x),  xm = bf16(x - xh),  xl = bf16(x - xh - xm)
// (round to nearest even, v_cvt_pk_bf16_f32; both subtractions are exact in f32).  Three 8-bit significands hold all
// 24 bits of an f32 and bf16 has f32's exponent range: the split is EXACT for every normal f32 whose third piece does
// not underflow (|x| > 2^-102) -- no scales, no bounds, no eligibility test.  A product then takes the six cross terms
// that matter,
//     W x = Wh xh + (Wh xm + Wm xh) + (Wm xm + Wh xl + Wl xh)  [+ Wm xl + Wl xm + Wl xl, dropped: <= 2^-23 |W x|]
// as six v_mfma_f32_16x16x32_bf16 with f32 accumulation (bf16 x bf16 products are exact in f32): 6 x 16 cycles per
// 16x16x32 block against 8 x 32 cycles of v_mfma_f32_16x16x4_f32 -- 2.7 x the f32 matrix rate with operands that are
// not narrower than f32.  The dropped terms are of the size of ONE f32 rounding of the running sum, which the f32 kernels
// commit at every one of their K additions (tests: against float64 on heavy-tailed weights and on rows spanning 40
// binades this path is at least as close as the f32 kernels).
//
// Only the dense layers run here (the five GEMMs, 60 % of the f32 network's time); the GRU recurrences stay on the
// f32 matrix cores (gru_rec3_kernel): their h fragments in three pieces would be 156 VGPRs per wavefront, which rules
// out the 192-sequence workgroup, and a 117 KB weight slab per unit tile makes them LDS-DMA-bound at no gain over the
// f32 kernel.  gru_rec3_kernel<..., true> writes h a second time as three-piece fragments for
// the GEMM that reads it.
//
// Layouts are kernels_h3.hip's with a third piece: row tiles of 16 SEQUENCES at one time step (row tile R = group * T
// + t); "TS3" [R][K-step S][h, m, l][64 lanes][8 bf16] = 3 KB per K-step, the fragments exactly as the MFMA takes them,
// with the same k-slot permutation inside a K-step (slot (q, j < 4) = unit tile 2 S, j >= 4 = unit tile 2 S + 1);
// weight fragment blocks [column block][S][tile][h, m, l][64][8] (tables_weights.cpp pack_panel_b3).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.h"
#include "nn_device.h"

typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
#define MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

namespace {

template <int T, int N> struct BFor {
    template <class F> static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, T>{});
        BFor<T + 1, N>::run(f);
    }
};
template <int N> struct BFor<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&&) {}
};

template <int OFF> __device__ __forceinline__ void lds_read_b128(f32x4& dst, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait3(f32x4& a, f32x4& b, f32x4& c)
{
    asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a), "+v"(b), "+v"(c) : "n"(N));
}

__device__ __forceinline__ float act_sigmoid_b(float x) { return 1.0f / (1.0f + expf(-x)); }

// two float4s (k-slots j = 0..3 and 4..7 of this lane) -> the three bf16 fragments
__device__ __forceinline__ void split8_3(const f32x4& a, const f32x4& b, b16x8& h, b16x8& m, b16x8& l)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float va = a[j], vb = b[j];
        const __bf16 ha = (__bf16)va, hb = (__bf16)vb;
        const float ra = va - (float)ha, rb = vb - (float)hb;
        const __bf16 ma = (__bf16)ra, mb = (__bf16)rb;
        h[j] = ha; h[4 + j] = hb;
        m[j] = ma; m[4 + j] = mb;
        l[j] = (__bf16)(ra - (float)ma);
        l[4 + j] = (__bf16)(rb - (float)mb);
    }
}

} // namespace

// ------------------------------------------------------------------ panel GEMM, bf16x3 (persistent)
// panel_gemm_h3_kernel's skeleton (one workgroup per CU walks (row panel of WAVES * RT row tiles, column block of NT
// tiles) items; weight fragment blocks stream through two LDS slabs by LDS-DMA, SP K-steps per phase; fragment reads
// are inline ds_read_b128 with counted lgkmcnt waits, a ring of D tiles ahead) with three pieces per operand and six
// MFMAs per (tile, row tile, K-step).
//   IN_TS: A in TS3 (a_ld = K-steps per row tile) or row-major f32 [sequence][seq_T][a_ld] (the features: gathered
//   and split in the loop).  OUT 0: row-major f32 [sequence][seq_T][c_ld] through the lane transpose (gi for
//   gru_rec3_kernel, tile-major unit order; the gains); OUT 2: TS3 (c_ld K-steps per row tile).
//   map_T / map_skip: output row tile Ro reads input row tile (Ro / (map_T - map_skip)) * map_T + map_skip +
//   Ro % (map_T - map_skip)  (fc2 skips the 4 warm-up steps).
template <int NT, int RT, int ACT, int SP, int D, int WAVES, bool IN_TS, int OUT>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void panel_gemm_b3_kernel(
    const float* __restrict__ A, int a_ld, const float* __restrict__ Wfrag,
    const float* __restrict__ bias, float* __restrict__ C, int c_ld, int seq_T, int S_steps, int k_tiles, int n_blocks,
    int n_valid_tiles, int map_T, int map_skip, unsigned n_items)
{
    static_assert(NT % D == 0, "ring slots must be compile-time");
    static_assert(OUT != 2 || NT % 2 == 0, "the split output pairs unit tiles");
    static_assert(NT * SP * 3072 <= 65536, "ds_read offset field");
    static_assert(3 * (D - 1) <= 15, "lgkmcnt field");
    __shared__ __attribute__((aligned(16))) float slab[2][NT * SP * 768];
    __shared__ __attribute__((aligned(16))) float sbias[2][NT * 16];
    typedef __attribute__((address_space(3))) float lds_float;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    const int P = (S_steps + SP - 1) / SP;

    // Operand positions are 32-bit float indices from A (every buffer is below 2^32 floats).  TS3: the row tile's part
    // is wave-uniform (a scalar), the lane adds 16 bytes times its number -- loads take the saddr + voffset form and no
    // per-lane 64-bit address is ever built (this kernel sits at the 256-register cap; a spilled register would be fatal,
    // not slow: the inline ds_reads and the LDS-DMA land in registers the compiler believes to be free to reuse).
    auto a_pointer = [&](unsigned item, int rt) -> unsigned {
        const unsigned panel = item / (unsigned)n_blocks;
        unsigned R = (panel * WAVES + wave) * RT + rt;
        if (map_T > 0) {
            const unsigned per = (unsigned)(map_T - map_skip);
            const unsigned g = R / per;
            R = g * (unsigned)map_T + (unsigned)map_skip + (R - g * per);
        }
        if (IN_TS) return R * (unsigned)a_ld * 768u; // scalar
        const unsigned g = R / (unsigned)seq_T, t = R - g * (unsigned)seq_T;
        return ((g * 16u + (unsigned)m) * (unsigned)seq_T + t) * (unsigned)a_ld + 4u * (unsigned)q;
    };
    auto issue = [&](unsigned item, int p, float* dst) {
        const int nblk = (int)(item % (unsigned)n_blocks);
        const int s0 = p * SP;
        const int cnt = (S_steps - s0 < SP) ? (S_steps - s0) : SP;
        const float* src = Wfrag + ((size_t)nblk * S_steps + s0) * (NT * 768);
        const int nb = NT * cnt * 3; // 1 KB blocks
        constexpr int PER = (NT * SP * 3 + WAVES - 1) / WAVES;
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
        if (tid < NT * 4) reinterpret_cast<f32x4*>(dst)[tid] = reinterpret_cast<const f32x4*>(bias + nblk * (NT * 16))[tid];
    };

    unsigned item = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    if (item >= n_items) return;
    const unsigned slab_addr[2] = {(unsigned)(uintptr_t)(lds_float*)slab[0] + (unsigned)lane * 16u,
                                   (unsigned)(uintptr_t)(lds_float*)slab[1] + (unsigned)lane * 16u};
    typedef const __attribute__((address_space(1))) f32x4* gptr4;
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};
    // one K-step of this lane's operand: TS3: the three fragments as stored; row-major: two float4s to be split
    typedef const __attribute__((address_space(1))) char* gbytes;
    unsigned lane16 = (unsigned)lane * 16u;
    auto load_step = [&](unsigned off, int s, f32x4& a4, f32x4& b4, f32x4& c4) {
        if (IN_TS) {
            // scalar base; the offset is UNSIGNED (readfirstlane returns an int: fc3's operand passes 2^31 floats at 49152 chunks)
            gbytes sb = (gbytes)(A + (size_t)(unsigned)__builtin_amdgcn_readfirstlane((int)off)) + 3072 * s;
            unsigned vo = lane16;
            asm volatile("" : "+v"(vo)); // keep the lane offset a 32-bit VGPR instead of a hoisted 64-bit address
            a4 = *(const __attribute__((address_space(1))) f32x4*)(sb + vo);
            b4 = *(const __attribute__((address_space(1))) f32x4*)(sb + 1024 + vo);
            c4 = *(const __attribute__((address_space(1))) f32x4*)(sb + 2048 + vo);
        } else {
            const float* base = A + off;
            a4 = *(gptr4)(base + 32 * s);
            b4 = (2 * s + 1 < k_tiles) ? *(gptr4)(base + 32 * s + 16) : zero4;
        }
    };
    auto to_frags = [&](const f32x4& a4, const f32x4& b4, const f32x4& c4, b16x8& h, b16x8& mm, b16x8& l) {
        if (IN_TS) { h = __builtin_bit_cast(b16x8, a4); mm = __builtin_bit_cast(b16x8, b4); l = __builtin_bit_cast(b16x8, c4); }
        else split8_3(a4, b4, h, mm, l);
    };

    unsigned a_ptr[RT];
    b16x8 xh[RT], xm[RT], xl[RT];       // this K-step's activation fragments
    f32x4 ra[RT], rb[RT], rc[RT];       // the K-step after it, as loaded
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        a_ptr[rt] = a_pointer(item, rt);
        f32x4 a0, b0, c0 = zero4;
        load_step(a_ptr[rt], 0, a0, b0, c0);
        to_frags(a0, b0, c0, xh[rt], xm[rt], xl[rt]);
        rc[rt] = zero4;
        load_step(a_ptr[rt], 1, ra[rt], rb[rt], rc[rt]);
    }
    issue(item, 0, slab[0]);
    stage_bias(item, sbias[0]);
    __syncthreads();
    int buf = 0, bbuf = 0;

    for (; item < n_items; item += gridDim.x) {
        const unsigned next_item = item + gridDim.x;
        const bool has_next = next_item < n_items;
        // accumulators start from zero and the bias is added in the epilogue (like MatMul + Add): starting both row
        // tiles' accumulators from the same bias registers makes the compiler keep copies of all of them alive across
        // the item loop -- 9 to 11 spilled registers in two of the four instances, and a spill is fatal here (a_pointer)
        f32x4 acc[RT][NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) acc[rt][t] = zero4;
        }
        unsigned a_nextitem[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) a_nextitem[rt] = has_next ? a_pointer(next_item, rt) : a_ptr[rt];

        for (int p = 0; p < P; ++p) {
            const int s0 = p * SP;
            const int cnt = (S_steps - s0 < SP) ? (S_steps - s0) : SP;
            if (p + 1 < P) issue(item, p + 1, slab[buf ^ 1]);
            else if (has_next) { issue(next_item, 0, slab[buf ^ 1]); stage_bias(next_item, sbias[bbuf ^ 1]); }

            unsigned rd = slab_addr[buf];
            f32x4 wh[D], wm[D], wl[D];
            BFor<0, D>::run([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                lds_read_b128<t * 3072>(wh[t], rd);
                lds_read_b128<t * 3072 + 1024>(wm[t], rd);
                lds_read_b128<t * 3072 + 2048>(wl[t], rd);
            });
            // one K-step; LAST: the last step of the phase -- the tiles past the ring's reach are in the other buffer,
            // behind the barrier, so nothing is read ahead for them and the waits count down instead (compile-time
            // either way: no read is ever in flight when the phase ends, and no control flow inside the tile loop)
            auto k_step = [&](int s, auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
                const int sg = s0 + s;
                BFor<0, NT>::run([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    constexpr int k = t % D;
                    constexpr int younger = LAST ? ((NT - 1 - t < D - 1) ? NT - 1 - t : D - 1) : D - 1;
                    lds_wait3<3 * younger>(wh[k], wm[k], wl[k]);
                    const b16x8 ah = __builtin_bit_cast(b16x8, wh[k]);
                    const b16x8 am = __builtin_bit_cast(b16x8, wm[k]);
                    const b16x8 al = __builtin_bit_cast(b16x8, wl[k]);
                    // smallest terms first
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_B(al, xh[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_B(ah, xl[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_B(am, xm[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_B(am, xh[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_B(ah, xm[rt], acc[rt][t]);
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA_B(ah, xh[rt], acc[rt][t]);
                    if constexpr (!LAST || t + D < NT) {
                        lds_read_b128<(t + D) * 3072>(wh[k], rd);
                        lds_read_b128<(t + D) * 3072 + 1024>(wm[k], rd);
                        lds_read_b128<(t + D) * 3072 + 2048>(wl[k], rd);
                    }
                });
                // next K-step's fragments from the values requested one step ago, then the request for the step
                // after it (the next item's first steps near the end: in flight across the epilogue)
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    to_frags(ra[rt], rb[rt], rc[rt], xh[rt], xm[rt], xl[rt]);
                    const int s2 = sg + 2;
                    if (s2 < S_steps) load_step(a_ptr[rt], s2, ra[rt], rb[rt], rc[rt]);
                    else load_step(a_nextitem[rt], s2 - S_steps, ra[rt], rb[rt], rc[rt]);
                }
                rd += NT * 3072;
            };
            for (int s = 0; s + 1 < cnt; ++s) k_step(s, std::false_type{});
            k_step(cnt - 1, std::true_type{});
            __syncthreads();
            buf ^= 1;
        }

        const int nblk = (int)(item % (unsigned)n_blocks);
        const unsigned panel = item / (unsigned)n_blocks;
        const int valid_t = n_valid_tiles - nblk * NT;
        const f32x4* bl = reinterpret_cast<const f32x4*>(sbias[bbuf]) + q;
        auto activate = [&](f32x4 v, int t) {
            v += bl[4 * t];
            if (ACT == FVAD_ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (ACT == FVAD_ACT_SIGMOID) {
                v.x = act_sigmoid_b(v.x); v.y = act_sigmoid_b(v.y); v.z = act_sigmoid_b(v.z); v.w = act_sigmoid_b(v.w);
            }
            return v;
        };
        float* c_ptr[RT];
        const int bp_addr = ((lane >> 2) + 16 * (lane & 3)) * 4; // row-major output: lane 4 m + q takes row m, columns 4 q ..
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned R = (panel * WAVES + wave) * RT + rt;
            if (OUT == 2) {
                c_ptr[rt] = C + ((size_t)R * (size_t)c_ld + (size_t)(nblk * NT / 2)) * 768 + lane * 4;
            } else {
                const unsigned g = R / (unsigned)seq_T, t = R - g * (unsigned)seq_T;
                c_ptr[rt] = C + ((size_t)(g * 16 + (unsigned)(lane >> 2)) * seq_T + t) * (size_t)c_ld + nblk * (NT * 16) + 4 * (lane & 3);
            }
        }
        auto emit = [&](int t, int rt) {
            const f32x4 v = activate(acc[rt][t], t);
            f32x4 o;
            o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[0])));
            o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[1])));
            o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[2])));
            o.w = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[3])));
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 16 * t) = o;
        };
        // split output: unit tiles 2 u and 2 u + 1 of this lane are the two halves of K-step u's fragments
        auto emit_pair = [&](int u, int rt) {
            const f32x4 v0 = activate(acc[rt][2 * u], 2 * u), v1 = activate(acc[rt][2 * u + 1], 2 * u + 1);
            b16x8 h, mm, l;
            split8_3(v0, v1, h, mm, l);
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 768 * u) = __builtin_bit_cast(f32x4, h);
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 768 * u + 256) = __builtin_bit_cast(f32x4, mm);
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 768 * u + 512) = __builtin_bit_cast(f32x4, l);
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
// in_ts: A is TS3 (a_ld = K-steps per row tile) or row-major f32 [sequence][seq_T][a_ld]; out: 0 row-major f32
// [sequence][seq_T][c_ld], 2 TS3 (c_ld K-steps per row tile).  K = true reduction length.  -1: no instance.
int fvad_launch_panel_gemm_b3(const float* A, int in_ts, int a_ld, const float* Wfrag, const float* bias, float* C,
                              int out, int c_ld, int seq_T, long row_tiles, int nt, int n_blocks, int K, int act,
                              int n_valid_tiles, int map_T, int map_skip, int n_wg, hipStream_t stream)
{
    if (row_tiles % 16) return -1;
    const int k_tiles = (K + 15) / 16;
    const int S_steps = (k_tiles + 1) / 2;
    if (S_steps < 3) return -1;
    // operand positions are 32-bit float indices (a_pointer)
    if (in_ts ? (double)row_tiles * a_ld * 768.0 >= 4294967296.0 : (double)row_tiles * 16.0 * a_ld >= 4294967296.0) return -1;
    const unsigned n_items = (unsigned)((row_tiles / 16) * n_blocks);
    const unsigned grid = n_items < (unsigned)n_wg ? n_items : (unsigned)n_wg;
#define CASEB(NT_, ACT_, SP_, D_, IN_, OUT_)                                                                  \
    if (nt == NT_ && act == ACT_ && (in_ts != 0) == IN_ && out == OUT_) {                                     \
        hipLaunchKernelGGL((panel_gemm_b3_kernel<NT_, 2, ACT_, SP_, D_, 8, IN_, OUT_>), dim3(grid), dim3(512), 0, \
                           stream, A, a_ld, Wfrag, bias, C, c_ld, seq_T, S_steps, k_tiles, n_blocks,          \
                           n_valid_tiles, map_T, map_skip, n_items);                                          \
        return 0;                                                                                             \
    }
    CASEB(15, FVAD_ACT_NONE, 1, 3, false, 0)   // features (row-major) -> gi (row-major, tile-major units)
    CASEB(15, FVAD_ACT_NONE, 1, 3, true, 0)    // h1 (TS3) -> gi
    CASEB(10, FVAD_ACT_RELU, 2, 5, true, 2)    // fc2, fc3 (TS3 -> TS3)
    CASEB(12, FVAD_ACT_SIGMOID, 1, 4, true, 0) // fc4 (TS3) -> gains (row-major)
#undef CASEB
    return -1;
}
