// kernels_nn.hip -- NSNet2 on gfx950 fp32 MFMA (v_mfma_f32_16x16x4_f32).
//
// Replaces onnx_instance.run() (reference src/NSNet2.zig:220) for a batch of independent
// 54-row sequences.  The GRU hidden state is zero at row 0 of every sequence (the reference's
// ORT session has no state tensors, NSNet2.zig:57-58,71-112), so sequences -- one per 0.5 s chunk
// of every stream -- are a pure batch axis; only the 54 steps inside a sequence are serial.
//
// Operand convention shared by both kernels ("row panel"): one wavefront owns 16 activation rows
// for the whole kernel.  Lane l = (m = l & 15, q = l >> 4).  For a 16-deep "super-step" S of the
// reduction dimension the lane holds the float4  Act[row m][16 S + 4 q .. + 3];  element r of it is
// the B operand of the r-th of four MFMAs.  Weights are pre-arranged on the host into "fragment
// blocks" of 64 lanes x 4 floats:  block(tile T, super-step S)[lane][r] = W[16 T + (lane & 15)]
// [16 S + 4 (lane >> 4) + r]  -- element r is the A operand of the r-th MFMA.  With weights as A
// and activations as B the result tile D[unit][row] lands as: lane (m, q) holds units
// 16 T + 4 q + {0..3} of row m in its 4 accumulator registers -- i.e. exactly the float4 the next
// layer (or the next GRU step) needs as its activation operand for super-step T.  No transposes,
// no LDS traffic for activations; LDS only stages weight fragment blocks, which are shared by the
// waves of a workgroup and read back with one conflict-free ds_read_b128 per lane per 4 MFMAs.
//
// Each output element is a k-ordered chain of f32 fmas (the MFMA is exact f32 fma, guide:
// MI355X_MICROARCH.md "Matrix cores"), bias added after the chain like MatMul+Add in the graph.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "kernels.h"

#include "nn_device.h"

__device__ __forceinline__ float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// ------------------------------------------------------------------ panel GEMM
// C[row][n0 + ...] = act(A[row][0..16 S_steps) . W^T + bias), one workgroup = WAVES*16 rows x
// one block of NT*16 output units.  grid = (rows / (16 WAVES), n_blocks).
// Wfrag: [n_blocks][S_steps][NT][64][4] floats.
template <int NT, int ACT, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void panel_gemm_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ Wfrag,
    const float* __restrict__ bias, float* __restrict__ C, int ldc, int S_steps, int row_map_T,
    int row_map_skip, int n_valid_tiles, const unsigned* guard)
{
    __shared__ __attribute__((aligned(16))) float slab[2][NT * 256];
    // launched behind gru_ws2_kernel as part of its fallback chain: runs only if that kernel raised *guard
    if (guard && *guard == 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m = lane & 15;
    const int q = lane >> 4;
    const int nblk = blockIdx.y;

    // Row mapping: when row_map_T > 0 the kernel's compact row index i addresses only rows
    // skip..T-1 of every T-row sequence of A (used to run fc2..fc4 on rows 4..53 only).
    const unsigned row = (blockIdx.x * WAVES + wave) * 16 + m;
    unsigned a_row = row;
    if (row_map_T > 0) {
        const unsigned per = (unsigned)(row_map_T - row_map_skip);
        const unsigned qd = row / per;
        a_row = qd * (unsigned)row_map_T + (unsigned)row_map_skip + (row - qd * per);
    }
    const float* a_ptr = A + (size_t)a_row * (size_t)lda + 4 * q;
    const float* w_src = Wfrag + (size_t)nblk * S_steps * (NT * 256);

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int SLAB_F4 = NT * 64;               // float4s per slab
    constexpr int PER_T = (SLAB_F4 + WAVES * 64 - 1) / (WAVES * 64);

    // prologue: slab 0 -> LDS
    {
        const f32x4* src = reinterpret_cast<const f32x4*>(w_src);
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * WAVES * 64;
            if (idx < SLAB_F4) reinterpret_cast<f32x4*>(slab[0])[idx] = src[idx];
        }
    }
    f32x4 a_cur = *reinterpret_cast<const f32x4*>(a_ptr);
    __syncthreads();

    for (int S = 0; S < S_steps; ++S) {
        const int cur = S & 1;
        f32x4 stage[PER_T];
        f32x4 a_next = a_cur;
        const bool more = (S + 1 < S_steps);
        if (more) {
            const f32x4* src = reinterpret_cast<const f32x4*>(w_src + (size_t)(S + 1) * (NT * 256));
#pragma unroll
            for (int i = 0; i < PER_T; ++i) {
                const int idx = tid + i * WAVES * 64;
                if (idx < SLAB_F4) stage[i] = src[idx];
            }
            a_next = *reinterpret_cast<const f32x4*>(a_ptr + 16 * (S + 1));
        }
        const f32x4* wl = reinterpret_cast<const f32x4*>(slab[cur]) + lane;
        // consecutive MFMAs go to different accumulators: a dependent v_mfma_f32_16x16x4_f32 issues every 40 cycles,
        // an independent one every 32 (each accumulator still sees its k in order)
        constexpr int TG = NT <= 12 ? NT : 5; // tiles per interleaved group (register budget); instances: 8, 11, 25
        static_assert(NT % TG == 0, "tile groups");
#pragma unroll
        for (int t0 = 0; t0 < NT; t0 += TG) {
            f32x4 w4[TG];
#pragma unroll
            for (int t = 0; t < TG; ++t) w4[t] = wl[(t0 + t) * 64];
#pragma unroll
            for (int t = 0; t < TG; ++t) acc[t0 + t] = MFMA16(w4[t].x, a_cur.x, acc[t0 + t]);
#pragma unroll
            for (int t = 0; t < TG; ++t) acc[t0 + t] = MFMA16(w4[t].y, a_cur.y, acc[t0 + t]);
#pragma unroll
            for (int t = 0; t < TG; ++t) acc[t0 + t] = MFMA16(w4[t].z, a_cur.z, acc[t0 + t]);
#pragma unroll
            for (int t = 0; t < TG; ++t) acc[t0 + t] = MFMA16(w4[t].w, a_cur.w, acc[t0 + t]);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < PER_T; ++i) {
                const int idx = tid + i * WAVES * 64;
                if (idx < SLAB_F4) reinterpret_cast<f32x4*>(slab[cur ^ 1])[idx] = stage[i];
            }
        }
        a_cur = a_next;
        __syncthreads();
    }

    float* c_ptr = C + (size_t)row * (size_t)ldc + nblk * (NT * 16) + 4 * q;
    const float* b_ptr = bias + nblk * (NT * 16) + 4 * q;
    const int valid_t = n_valid_tiles - nblk * NT; // tiles past the output's width are computed but not stored
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < valid_t) { // (no `break`: the loop must unroll completely, or the accumulators end up in scratch)
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b_ptr + 16 * t);
            f32x4 v = acc[t] + b4;
            if (ACT == FVAD_ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (ACT == FVAD_ACT_SIGMOID) {
                v.x = act_sigmoid(v.x); v.y = act_sigmoid(v.y); v.z = act_sigmoid(v.z); v.w = act_sigmoid(v.w);
            }
            *reinterpret_cast<f32x4*>(c_ptr + 16 * t) = v;
        }
    }
}

template <int NT, int ACT>
static void launch_panel(const float* A, int lda, const float* Wfrag, const float* bias, float* C,
                         int ldc, long rows, int n_blocks, int S_steps, int map_T, int map_skip,
                         int n_valid_tiles, const unsigned* guard, hipStream_t stream)
{
    constexpr int WAVES = 4;
    dim3 grid((unsigned)(rows / (16 * WAVES)), (unsigned)n_blocks);
    hipLaunchKernelGGL((panel_gemm_kernel<NT, ACT, WAVES>), grid, dim3(WAVES * 64), 0, stream, A,
                       lda, Wfrag, bias, C, ldc, S_steps, map_T, map_skip, n_valid_tiles, guard);
}

// rows must be a multiple of 64; NT selects the per-block width (16*NT units); n_valid_tiles <= 0: every tile of
// every block is stored; guard != nullptr: the kernel returns at once unless *guard != 0.
int fvad_launch_panel_gemm(const float* A, int lda, const float* Wfrag, const float* bias, float* C,
                           int ldc, long rows, int nt, int n_blocks, int S_steps, int act,
                           int map_T, int map_skip, hipStream_t stream, int n_valid_tiles, const unsigned* guard)
{
    if (n_valid_tiles <= 0) n_valid_tiles = nt * n_blocks;
    if (rows <= 0 || rows % 64 != 0) return -1; // whole 64-row workgroups only (the engine pads the batch)
#define CASE(NT_, ACT_)                                                                          \
    if (nt == NT_ && act == ACT_) {                                                              \
        launch_panel<NT_, ACT_>(A, lda, Wfrag, bias, C, ldc, rows, n_blocks, S_steps, map_T,     \
                                map_skip, n_valid_tiles, guard, stream);                         \
        return 0;                                                                                \
    }
    CASE(25, FVAD_ACT_NONE)
    CASE(11, FVAD_ACT_SIGMOID)
    // models of other dimensions than NSNet2-baseline's (run_nn_generic): any width as blocks of 8 tiles
    CASE(8, FVAD_ACT_NONE)
    CASE(8, FVAD_ACT_RELU)
#undef CASE
    return -1;
}

// ------------------------------------------------------------------ panel GEMM, small batches
// The same 64-row x (16 NT)-column workgroup for launches of a few to a few thousand rows (a push of one chunk,
// BASELINE config 3's 82), where a layer is a handful of row panels and what is exposed is the latency chain of one
// workgroup, not the MFMA rate.  Against panel_gemm_kernel:
//   * the reduction length S (super-steps of 16) is a template parameter and the phase loop is unrolled completely,
//     so no load sits under a run-time predicate: with `if (more)` around the next slab's loads the compiler put
//     s_waitcnt vmcnt(0) in the middle of the MFMA section and serialised the prologue's loads one by one;
//   * slab and activation loads run PF phases (of KS super-steps) ahead through registers -- a first-touch
//     activation load takes longer than one phase's MFMAs -- and LDS holds PF + 1 slabs, one barrier per phase;
//   * narrow column blocks (2 or 4 tiles): 4800 rows x 600 columns are 750 (NT = 4) workgroups instead of 600, and
//     each wavefront's chain is shorter.
// tools/small_gemm.hip measures the variants: fc2 / fc3 / fc4 of the 82-chunk batch 41.7 / 59.3 / 33.9 us ->
// 24.3 / 34.2 / 14.2 us, of a one-chunk push 25 -> 8 us.  Every output is still the k-ordered chain of f32 fmas with
// the bias added after it: bit-identical to panel_gemm_kernel (asserted by the tool and by the parity tests).
// Rows: the launch covers `rm.n_rows` compact rows r (the grid is ceil(n_rows / 64) panels; a panel's rows past n_rows read
// row n_rows - 1 again and store nothing).  rm.L > 0: r = seq * L + tt is steps [t0, t0 + L) of sequence seq -- A's row
// seq * in_T + in_t0 + tt, C's row seq * out_T + out_t0 + tt -- which serves the warm-up skip (L = T - skip, in_T = T,
// in_t0 = skip, out rows compact) and launches over the REAL sequences of a padded batch (n_rows = n_real * L).  rm.L == 0:
// rows as they are.  A row's arithmetic does not depend on which panel or launch computes it.
template <int NT, int KS, int S, int PF, int ACT>
__global__ __launch_bounds__(256) void panel_gemm_s_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ Wfrag, const float* __restrict__ bias,
    float* __restrict__ C, int ldc, GemmRowMap rm, int n_valid_tiles, const unsigned* guard)
{
    constexpr int NB = PF + 1;
    __shared__ __attribute__((aligned(16))) float slab[NB][KS * NT * 256];
    // launched behind gru_ws2_kernel as part of its fallback chain: runs only if that kernel raised *guard
    if (guard && *guard == 0) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int m = lane & 15;
    const int q = lane >> 4;
    const int nblk = blockIdx.y;
    const unsigned row_raw = (blockIdx.x * 4 + wave) * 16 + m;
    const bool row_ok = row_raw < (unsigned)rm.n_rows;
    const unsigned row = row_ok ? row_raw : (unsigned)rm.n_rows - 1u;
    unsigned a_row = row, c_row = row;
    if (rm.L > 0) {
        const unsigned qd = row / (unsigned)rm.L, tt = row - qd * (unsigned)rm.L;
        a_row = qd * (unsigned)rm.in_T + (unsigned)rm.in_t0 + tt;
        c_row = qd * (unsigned)rm.out_T + (unsigned)rm.out_t0 + tt;
    }
    const float* a_ptr = A + (size_t)a_row * (size_t)lda + 4 * q;
    const f32x4* w_src = reinterpret_cast<const f32x4*>(Wfrag + (size_t)nblk * S * (NT * 256));

    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NP = (S + KS - 1) / KS;              // phases; the last one may be short
    constexpr int PER_T = (KS * NT * 64 + 255) / 256;  // float4s of a slab per thread
    f32x4 stage[NB][PER_T];
    f32x4 a_q[NB][KS];
    auto ks_of = [](int p) { return (S - p * KS) < KS ? (S - p * KS) : KS; };
    auto load = [&](int p) { // phase p: slab -> stage registers, activations -> a_q (p is a constant after unrolling)
        const int ks = ks_of(p);
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * 256;
            if ((i + 1) * 256 <= ks * NT * 64 || idx < ks * NT * 64) stage[p % NB][i] = w_src[(size_t)p * KS * NT * 64 + idx];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (s < ks) a_q[p % NB][s] = *reinterpret_cast<const f32x4*>(a_ptr + 16 * (p * KS + s));
    };
    auto to_lds = [&](int p) {
        const int ks = ks_of(p);
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * 256;
            if ((i + 1) * 256 <= ks * NT * 64 || idx < ks * NT * 64) reinterpret_cast<f32x4*>(slab[p % NB])[idx] = stage[p % NB][i];
        }
    };
#pragma unroll
    for (int p = 0; p < PF && p < NP; ++p) load(p);
    to_lds(0);
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        if (p + PF < NP) load(p + PF);
        const int ks = ks_of(p);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s < ks) {
                const f32x4* wl = reinterpret_cast<const f32x4*>(slab[p % NB]) + s * NT * 64 + lane;
                f32x4 w4[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w4[t] = wl[t * 64];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].x, a_q[p % NB][s].x, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].y, a_q[p % NB][s].y, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].z, a_q[p % NB][s].z, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].w, a_q[p % NB][s].w, acc[t]);
            }
        }
        if (p + 1 < NP) {
            // slab p + 1 goes into the buffer last read in phase p + 1 - NB <= p - 1: every wavefront has passed the
            // barrier that ended that phase
            to_lds(p + 1);
            __syncthreads();
        }
    }

    float* c_ptr = C + (size_t)c_row * (size_t)ldc + nblk * (NT * 16) + 4 * q;
    const float* b_ptr = bias + nblk * (NT * 16) + 4 * q;
    const int valid_t = row_ok ? n_valid_tiles - nblk * NT : 0; // tiles past the output's width are computed but not stored
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < valid_t) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b_ptr + 16 * t);
            f32x4 v = acc[t] + b4;
            if (ACT == FVAD_ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (ACT == FVAD_ACT_SIGMOID) {
                v.x = act_sigmoid(v.x); v.y = act_sigmoid(v.y); v.z = act_sigmoid(v.z); v.w = act_sigmoid(v.w);
            }
            *reinterpret_cast<f32x4*>(c_ptr + 16 * t) = v;
        }
    }
}

// rows must be a multiple of 64; Wfrag from pack_panel with blocks of nt tiles (nt = 2: launches of up to ~2000 rows,
// nt = 4: larger ones); the (S_steps, act) pairs of NSNet2-baseline's five layers; n_valid_tiles <= 0: every tile of
// every block is stored; guard != nullptr: the kernel returns at once unless *guard != 0.
int fvad_launch_panel_gemm_s(const float* A, int lda, const float* Wfrag, const float* bias, float* C, int ldc,
                             long rows, int nt, int n_blocks, int S_steps, int act, int map_T, int map_skip,
                             hipStream_t stream, int n_valid_tiles, const unsigned* guard)
{
    if (rows <= 0 || rows % 64 != 0) return -1; // (this form: whole 64-row panels of a padded batch)
    GemmRowMap rm{};
    rm.n_rows = (int)rows;
    if (map_T > 0) { rm.L = map_T - map_skip; rm.in_T = map_T; rm.in_t0 = map_skip; rm.out_T = rm.L; rm.out_t0 = 0; }
    return fvad_launch_panel_gemm_s_rows(A, lda, Wfrag, bias, C, ldc, rm, nt, n_blocks, S_steps, act, stream, n_valid_tiles, guard);
}

int fvad_launch_panel_gemm_s_rows(const float* A, int lda, const float* Wfrag, const float* bias, float* C, int ldc,
                                  GemmRowMap rm, int nt, int n_blocks, int S_steps, int act, hipStream_t stream,
                                  int n_valid_tiles, const unsigned* guard)
{
    if (n_valid_tiles <= 0) n_valid_tiles = nt * n_blocks;
    if (rm.n_rows <= 0 || rm.L < 0 || (rm.L > 0 && (rm.in_T < rm.L || rm.out_T < rm.L || rm.in_t0 < 0 || rm.out_t0 < 0))) return -1;
    const dim3 grid((unsigned)((rm.n_rows + 63) / 64), (unsigned)n_blocks);
#define CASES(S_, ACT_)                                                                                              \
    if (S_steps == S_ && act == ACT_) {                                                                              \
        if (nt == 2)                                                                                                 \
            hipLaunchKernelGGL((panel_gemm_s_kernel<2, 2, S_, 3, ACT_>), grid, dim3(256), 0, stream, A, lda, Wfrag,  \
                               bias, C, ldc, rm, n_valid_tiles, guard);                                              \
        else if (nt == 4)                                                                                            \
            hipLaunchKernelGGL((panel_gemm_s_kernel<4, 1, S_, 4, ACT_>), grid, dim3(256), 0, stream, A, lda, Wfrag,  \
                               bias, C, ldc, rm, n_valid_tiles, guard);                                              \
        else                                                                                                         \
            return -1;                                                                                               \
        return 0;                                                                                                    \
    }
    CASES(11, FVAD_ACT_NONE)    // features -> gi (fc1 folded): K = 161 -> 176
    CASES(25, FVAD_ACT_NONE)    // h1 -> gi
    CASES(25, FVAD_ACT_RELU)    // fc2
    CASES(38, FVAD_ACT_RELU)    // fc3: K = 600 -> 608
    CASES(38, FVAD_ACT_SIGMOID) // fc4
#undef CASES
    return -1;
}

// ------------------------------------------------------------------ panel GEMM, v3 (persistent)
// Device timestamps (s_memtime) of a plain double-buffered LDS-DMA version showed where its missing ~20 % went:
//   * the compiler guards every ds_read behind s_waitcnt vmcnt(0) while an LDS-DMA is in flight (it
//     cannot prove the DMA targets the *other* slab buffer), so the first super-step of each phase
//     stalled for the whole next-slab DMA and the first super-step after an epilogue stalled until
//     all 30 accumulator stores were acknowledged by memory;
//   * all NT fragment reads of a super-step were issued together and waited for with lgkmcnt(0),
//     which leaves a wavefront that runs alone on its SIMD (the two wavefronts of a SIMD are not
//     arbitrated fairly, one finishes a phase ~25 % early) at ~70 % of the MFMA rate.
// This kernel therefore
//   * is persistent: a workgroup walks items blockIdx.x, +gridDim.x, ... as one flat sequence of
//     phases, so the next item's first slab, first activations and bias block are in flight during
//     the current item's last phase;
//   * reads fragment blocks with inline ds_read_b128 + explicit lgkmcnt waits, one super-step ahead:
//     the read of tile t for step s+1 is issued right after the MFMAs of tile t in step s, so every
//     wait is lgkmcnt(NT-1) on data requested ~NT MFMA groups earlier; the compiler no longer sees
//     LDS reads that could alias the DMA;
//   * loads the activations of the next item's first two super-steps before the epilogue stores
//     (vmcnt retires in order on gfx9: a load issued after the stores cannot be waited for without
//     waiting for the stores too), so the stores have two super-steps to drain in the background.
template <int T, int N> struct StaticFor {
    template <class F> static __device__ __forceinline__ void run(F&& f) {
        f(std::integral_constant<int, T>{});
        StaticFor<T + 1, N>::run(f);
    }
};
template <int N> struct StaticFor<N, N> {
    template <class F> static __device__ __forceinline__ void run(F&&) {}
};

template <int OFF> __device__ __forceinline__ void lds_read_b128(f32x4& dst, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void lds_wait(f32x4& dst)
{
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(dst) : "n"(N));
}

template <int NT, int RT, int ACT, int SP, int WAVES>
__global__ __launch_bounds__(WAVES * 64) __attribute__((amdgpu_waves_per_eu(2, 2))) void panel_gemm3_kernel(
    const float* __restrict__ A, int lda, const float* __restrict__ Wfrag,
    const float* __restrict__ bias, float* __restrict__ C, int ldc, int S_steps, int n_blocks,
    int n_valid_tiles, int row_map_T, int row_map_skip, unsigned n_items, int tail_r)
{
    __shared__ __attribute__((aligned(16))) float slab[2][NT * SP * 256];
    __shared__ __attribute__((aligned(16))) float sbias[2][NT * 16];
    typedef __attribute__((address_space(3))) float lds_float;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6); // scalar: DMA addresses stay in SGPRs
    const int m = lane & 15;
    const int q = lane >> 4;
    const int P = (S_steps + SP - 1) / SP;
    constexpr int WAITN = (NT - 1 < 15) ? NT - 1 : 15; // lgkmcnt is a 4-bit field

    auto a_pointer = [&](unsigned item, int rt) -> const float* {
        const unsigned panel = item / (unsigned)n_blocks;
        const unsigned row = ((panel * WAVES + wave) * RT + rt) * 16 + m;
        unsigned a_row = row;
        if (row_map_T > 0) {
            const unsigned per = (unsigned)(row_map_T - row_map_skip);
            const unsigned qd = row / per;
            a_row = qd * (unsigned)row_map_T + (unsigned)row_map_skip + (row - qd * per);
        }
        return A + (size_t)a_row * (size_t)lda + 4 * q;
    };
    // every wavefront issues the same, compile-time number of DMA instructions per phase (surplus
    // ones repeat the last block), which lets the compiler count them in its vmcnt bookkeeping
    auto issue = [&](unsigned item, int p, float* dst) {
        const int nblk = (int)(item % (unsigned)n_blocks);
        const int s0 = p * SP;
        const int cnt = (S_steps - s0 < SP) ? (S_steps - s0) : SP;
        const float* src = Wfrag + ((size_t)nblk * S_steps + s0) * (NT * 256);
        const int nb = NT * cnt;
        constexpr int PER = (NT * SP + WAVES - 1) / WAVES;
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

    // XCD-aware start: workgroups are dealt to the 8 XCDs round-robin (blockIdx.x % 8), each with its own
    // L2.  Renumbering them XCD-major makes the workgroups of one XCD take consecutive items, i.e. the
    // column blocks of the same row panel, so a panel is fetched from HBM by one L2 instead of five.
    unsigned item = (gridDim.x % 8u == 0u) ? (blockIdx.x % 8u) * (gridDim.x / 8u) + blockIdx.x / 8u : blockIdx.x;
    if (item >= n_items) return;
    const unsigned slab_addr[2] = {(unsigned)(uintptr_t)(lds_float*)slab[0] + (unsigned)lane * 16u,
                                   (unsigned)(uintptr_t)(lds_float*)slab[1] + (unsigned)lane * 16u};
    typedef const __attribute__((address_space(1))) f32x4* gptr4; // keep these global_load, not flat_load
    const float* a_ptr[RT];
    f32x4 a0[RT], apre[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        a_ptr[rt] = a_pointer(item, rt);
        a0[rt] = *(gptr4)(a_ptr[rt]);
        apre[rt] = *(gptr4)(a_ptr[rt] + 16);
    }
    issue(item, 0, slab[0]);
    stage_bias(item, sbias[0]);
    __syncthreads();
    int buf = 0, bbuf = 0;

    for (; item < n_items; item += gridDim.x) {
        const unsigned next_item = item + gridDim.x;
        const bool has_next = next_item < n_items;
        // accumulators start from the bias: the epilogue competes with the SIMD partner's MFMAs for
        // VALU issue slots, so it should be little more than the stores
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
            f32x4 w[NT];
            StaticFor<0, NT>::run([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                lds_read_b128<t * 1024>(w[t], rd);
            });
            // one super-step; LAST: the last one of the phase
            auto super_step = [&](int s, auto last_c) {
                constexpr bool LAST = decltype(last_c)::value;
                const int sg = s0 + s;
                f32x4 a1[RT];
                if (sg == 0) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) a1[rt] = apre[rt]; // requested before the previous epilogue
                } else {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt)
                        a1[rt] = *(gptr4)((sg + 1 < S_steps) ? a_ptr[rt] + 16 * (sg + 1) : a_nextitem[rt]);
                }
                // MFMA r of a super-step reduces over k = 16 sg + 4q + r: when K is not a multiple of 16 the
                // last super-step's higher r are all padding (K = 161: only r = 0 carries k = 160)
                const int nr = (sg + 1 < S_steps) ? 4 : tail_r;
                auto mfmas = [&](auto tc) {
                    constexpr int t = decltype(tc)::value;
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA16(w[t].x, a0[rt].x, acc[rt][t]);
                    if (nr > 1) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA16(w[t].y, a0[rt].y, acc[rt][t]);
                    }
                    if (nr > 2) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA16(w[t].z, a0[rt].z, acc[rt][t]);
                    }
                    if (nr > 3) {
#pragma unroll
                        for (int rt = 0; rt < RT; ++rt) acc[rt][t] = MFMA16(w[t].w, a0[rt].w, acc[rt][t]);
                    }
                };
                // The read of tile t for the next super-step follows tile t's MFMAs, so every wait below sees exactly
                // NT-1 younger reads in flight.  In the last step of a phase the next data is not in LDS yet (other
                // buffer, after the barrier): nothing is read ahead and the waits count down -- no read may still be in
                // flight when the phase ends, because the compiler does not know about these reads and reuses their
                // registers at once (round 3: a late read landing in an epilogue address register faulted the bf16x3
                // form of this loop; here it had never shown, by timing only).
                StaticFor<0, NT>::run([&](auto tc) {
                    constexpr int t = decltype(tc)::value;
                    if constexpr (LAST) {
                        lds_wait<((NT - 1 - t) < 15 ? (NT - 1 - t) : 15)>(w[t]);
                        mfmas(tc);
                    } else {
                        lds_wait<WAITN>(w[t]);
                        mfmas(tc);
                        lds_read_b128<(NT + t) * 1024>(w[t], rd);
                    }
                });
                rd += NT * 1024;
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) a0[rt] = a1[rt];
            };
            for (int s = 0; s + 1 < cnt; ++s) super_step(s, std::false_type{});
            super_step(cnt - 1, std::true_type{});
            __syncthreads(); // next phase's weights (and bias block) landed; this buffer is free again
            buf ^= 1;
        }

        // epilogue of this item; the next item's first slab is already in LDS
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) apre[rt] = *(gptr4)(a_nextitem[rt] + 16);
        const int nblk = (int)(item % (unsigned)n_blocks);
        const unsigned panel = item / (unsigned)n_blocks;
        // The store path merges adjacent lanes only: in the MFMA layout (lane = m + 16 q) adjacent lanes
        // are different rows and a float4 store becomes 64 separate 16-byte requests (18 B/clk per CU,
        // tools/store_rate.hip).  A 16x4 lane transpose through the LDS crossbar (ds_bpermute, no LDS
        // memory) makes lane 4m + q hold row m's 16 bytes at column 4q, so four adjacent lanes write
        // 64 contiguous bytes and the same bytes drain at 50 B/clk.
        const int bp_addr = ((lane >> 2) + 16 * (lane & 3)) * 4; // this lane's data comes from lane m + 16 q
        float* c_ptr[RT];
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const unsigned row = ((panel * WAVES + wave) * RT + rt) * 16 + (unsigned)(lane >> 2);
            c_ptr[rt] = C + (size_t)row * (size_t)ldc + nblk * (NT * 16) + 4 * (lane & 3);
        }
        const int valid_t = n_valid_tiles - nblk * NT;
        auto emit = [&](int t, int rt) {
            f32x4 v = acc[rt][t];
            if (ACT == FVAD_ACT_RELU) {
                v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            } else if (ACT == FVAD_ACT_SIGMOID) {
                v.x = act_sigmoid(v.x); v.y = act_sigmoid(v.y); v.z = act_sigmoid(v.z); v.w = act_sigmoid(v.w);
            }
            f32x4 o;
            o.x = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[0])));
            o.y = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[1])));
            o.z = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[2])));
            o.w = __int_as_float(__builtin_amdgcn_ds_bpermute(bp_addr, __float_as_int(v[3])));
            *reinterpret_cast<f32x4*>(c_ptr[rt] + 16 * t) = o;
        };
        if (valid_t >= NT) { // every tile is stored: no per-tile branches, the permutes of many tiles overlap
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

// rows must be a multiple of 256; grid = one persistent workgroup per CU.  Returns -1 when there is no
// instance for (nt, act).
int fvad_launch_panel_gemm3(const float* A, int lda, const float* Wfrag, const float* bias, float* C,
                            int ldc, long rows, int nt, int n_blocks, int S_steps, int K, int act,
                            int n_valid_tiles, int map_T, int map_skip, int n_wg, hipStream_t stream)
{
    if (rows % 256) return -1;
    const int n_last = K - 16 * (S_steps - 1); // valid k in the last super-step
    const int tail_r = n_last >= 4 ? 4 : (n_last < 1 ? 4 : n_last);
    const unsigned n_items = (unsigned)((rows / 256) * n_blocks);
    const unsigned grid = n_items < (unsigned)n_wg ? n_items : (unsigned)n_wg;
#define CASE3(NT_, ACT_, SP_)                                                                         \
    if (nt == NT_ && act == ACT_) {                                                                   \
        hipLaunchKernelGGL((panel_gemm3_kernel<NT_, 2, ACT_, SP_, 8>), dim3(grid), dim3(512), 0, stream, \
                           A, lda, Wfrag, bias, C, ldc, S_steps, n_blocks, n_valid_tiles, map_T,      \
                           map_skip, n_items, tail_r);                                                \
        return 0;                                                                                     \
    }
    // no 19-tile instance: 152 accumulator + 76 fragment registers do not fit 256 VGPRs; fc2/fc3 use
    // three 13-tile column blocks here
    CASE3(15, FVAD_ACT_NONE, 5)
    CASE3(13, FVAD_ACT_RELU, 5)
    CASE3(11, FVAD_ACT_SIGMOID, 6)
#undef CASE3
    return -1;
}

// ------------------------------------------------------------------ GRU recurrence
// h_t = GRU(gi_t, h_{t-1}) for T steps, 16 sequences per wavefront.
//   gi   [n_seq_pad * T][3H]   = x_t W^T + Wb   (from the panel GEMMs), gate order z,r,h
//   R2frag [25 J][3 g][25 S][64][4]             recurrent weights as 1 KB fragment blocks
//   bR   [3H]
//   hout [n_seq_pad * T][H]
// Specialised for H = 400 (25 unit tiles of 16 units, 25 super-steps of 16).  h_{t-1} stays in
// registers as the 25 activation float4s (100 VGPRs); new h values are written straight to hout in the
// layout the lane itself re-reads as next step's operand, so the only cross-lane traffic in the
// recurrence is the MFMA itself.
// (GRU_H, GRU_J, GRU2_SLAB and the gate nonlinearities live in nn_device.h)

// ------------------------------------------------------------------ GRU recurrence, v3
// A plain LDS-DMA double-buffered recurrence with the two stalls its device timeline showed removed (same cure as panel_gemm3):
//   * fragment reads are inline ds_read_b128 with explicit lgkmcnt waits, two super-steps of (z, r, n)
//     blocks in flight, so the compiler no longer puts s_waitcnt vmcnt(0) (= "next slab's DMA has
//     landed") in front of the first LDS read of every unit tile;
//   * gi already contains Wb + Rb for the z and r gates (folded on the host), only the n gate's Rb is
//     added here: two loads and eight registers less per unit tile;
//   * gi is TILE-major here, gi[row][25 J][3 gates][16 units] (the host permutes the rows of the input
//     projection's weights, so the GEMM writes this order by itself): a unit tile's z, r, n operands are 192
//     contiguous bytes of the row.  In gate-major order they are three 64-byte segments 1600 bytes apart, each
//     in a 128-byte line whose other half belongs to the neighbouring tile, ~32k cycles away -- by then out
//     of L2, so every gi byte was fetched from HBM twice (tools/fetch_calib.hip + the PMC passes);
//   * the end-of-tile barrier waits with vmcnt(1): everything up to the slab DMA, but not the h store
//     issued just before it (__syncthreads() would wait for that store's acknowledgement too).
// wave must be wave-uniform (SGPR): the DMA then addresses global memory as scalar base + lane * 16
template <int WAVES>
__device__ __forceinline__ void gru3_issue_slab(const float* __restrict__ src, float* lds_dst, int wave, unsigned lane16)
{
    asm volatile("" : "+v"(lane16)); // keep the lane offset a 32-bit VGPR instead of a hoisted 64-bit address
#pragma unroll
    for (int b = wave; b < 3 * GRU_J; b += WAVES) {
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void*)((const __attribute__((address_space(1))) char*)src + b * 1024 + lane16),
            (__attribute__((address_space(3))) void*)(lds_dst + b * 256), 16, 0, 0);
    }
}

// TS3: h is written a second time as three bf16 pieces per value in the tiled fragment layout of kernels_b3.hip
// (hs3: [16-sequence group * T + t][13 K-steps][h, m, l][64 lanes][8 bf16]) for the bf16x3 GEMM that reads it; the
// recurrence itself keeps reading its own f32 rows
template <int WAVES, int D, bool TS3>
__global__ __launch_bounds__(WAVES * 64) void gru_rec3_kernel(const float* __restrict__ gi,
                                                              const float* __restrict__ R2frag,
                                                              const float* __restrict__ bR,
                                                              float* hout, int T, float* hs3)
{
    __shared__ __attribute__((aligned(16))) float slab[2][GRU2_SLAB];
    typedef __attribute__((address_space(3))) float lds_float;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    // Addresses are a scalar base (workgroup, wavefront, t, J) plus one 32-bit per-lane byte offset per
    // array, so that the loads take the saddr + voffset form and the 100 registers of h_{t-1} leave room
    // for three wavefronts per SIMD.
    typedef const __attribute__((address_space(1))) char* gbytes;
    const size_t seq0 = (size_t)(blockIdx.x * WAVES + wave) * 16;
    gbytes gi_w = (gbytes)(gi + seq0 * T * (3 * GRU_H));
    __attribute__((address_space(1))) char* h_w = (__attribute__((address_space(1))) char*)(hout + seq0 * T * GRU_H);
    gbytes bR_b = (gbytes)bR;
    const unsigned gi_off = ((unsigned)m * (unsigned)T * (3 * GRU_H) + 4u * q) * 4u;
    const unsigned h_off = ((unsigned)m * (unsigned)T * GRU_H + 4u * q) * 4u;
    const unsigned b_off = 16u * q;
    // the empty asm keeps the compiler from folding the lane offset into a hoisted 64-bit VGPR base
    auto ld4 = [](gbytes base, unsigned off) {
        asm volatile("" : "+v"(off));
        return *(const __attribute__((address_space(1))) f32x4*)(base + off);
    };
    const unsigned slab_addr[2] = {(unsigned)(uintptr_t)(lds_float*)slab[0] + (unsigned)lane * 16u,
                                   (unsigned)(uintptr_t)(lds_float*)slab[1] + (unsigned)lane * 16u};
    // unit tile J of h_t as three bf16 pieces (x = h + m + l exactly): half (J & 1) of K-step J / 2's fragments, 8
    // bytes per piece and lane; the last tile also writes the zero upper half of K-step 12
    __attribute__((address_space(1))) char* hs_w = nullptr;
    if (TS3) hs_w = (__attribute__((address_space(1))) char*)(hs3 + (size_t)(blockIdx.x * WAVES + wave) * T * (13 * 768));
    auto store_split3 = [&](__attribute__((address_space(1))) char* hs_t, int J, const f32x4& h) {
        typedef __bf16 b16x4 __attribute__((ext_vector_type(4)));
        typedef __bf16 b16x8 __attribute__((ext_vector_type(8)));
        b16x4 ph, pm, pl;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float v = h[r];
            const __bf16 a = (__bf16)v;
            const float ra = v - (float)a;
            const __bf16 b = (__bf16)ra;
            ph[r] = a; pm[r] = b; pl[r] = (__bf16)(ra - (float)b);
        }
        unsigned o = (unsigned)lane * 16u;
        asm volatile("" : "+v"(o));
        __attribute__((address_space(1))) char* dst = hs_t + (J >> 1) * 3072 + (J & 1) * 8 + o;
        if (J == GRU_J - 1) {
            const b16x4 z4 = (b16x4){0, 0, 0, 0};
            *(__attribute__((address_space(1))) b16x8*)dst = __builtin_shufflevector(ph, z4, 0, 1, 2, 3, 4, 5, 6, 7);
            *(__attribute__((address_space(1))) b16x8*)(dst + 1024) = __builtin_shufflevector(pm, z4, 0, 1, 2, 3, 4, 5, 6, 7);
            *(__attribute__((address_space(1))) b16x8*)(dst + 2048) = __builtin_shufflevector(pl, z4, 0, 1, 2, 3, 4, 5, 6, 7);
        } else {
            *(__attribute__((address_space(1))) b16x4*)dst = ph;
            *(__attribute__((address_space(1))) b16x4*)(dst + 1024) = pm;
            *(__attribute__((address_space(1))) b16x4*)(dst + 2048) = pl;
        }
    };

    gru3_issue_slab<WAVES>(R2frag, slab[0], wave, (unsigned)lane * 16u);

    // ---- t = 0: h_{-1} = 0, so R h + Rb = Rb
    for (int J = 0; J < GRU_J; ++J) {
        const f32x4 giz = ld4(gi_w + 192 * J, gi_off);
        const f32x4 gir = ld4(gi_w + 192 * J + 64, gi_off);
        const f32x4 gin = ld4(gi_w + 192 * J + 128, gi_off);
        const f32x4 bn = ld4(bR_b + 64 * J + 8 * GRU_H, b_off);
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = fast_sigmoid(giz[r]);
            const float rr = fast_sigmoid(gir[r]);
            const float n = fast_tanh(gin[r] + rr * bn[r]);
            h[r] = (1.0f - z) * n + z * 0.0f;
        }
        *(__attribute__((address_space(1))) f32x4*)(h_w + 64 * J + h_off) = h;
        if (TS3) store_split3(hs_w, J, h);
    }
    __syncthreads(); // drains the LDS-DMA (vmcnt) and publishes slab 0
    int buf = 0;

    for (int t = 1; t < T; ++t) {
        gbytes gi_t = gi_w + (size_t)t * (12 * GRU_H);
        gbytes h_prev = (gbytes)h_w + (size_t)(t - 1) * (4 * GRU_H);
        __attribute__((address_space(1))) char* h_out = h_w + (size_t)t * (4 * GRU_H);

        f32x4 hreg[GRU_J];
#pragma unroll
        for (int S = 0; S < GRU_J; ++S) hreg[S] = ld4(h_prev + 64 * S, h_off);
        // Consume h_{t-1} here, once per step: otherwise the compiler's wait for these loads sits in front
        // of the first MFMA of every unit tile as s_waitcnt vmcnt(0), i.e. behind that tile's slab DMA.
#pragma unroll
        for (int S = 0; S < GRU_J; ++S) asm volatile("" : "+v"(hreg[S]));

        for (int J = 0; J < GRU_J; ++J) {
            const int nJ = (J + 1 == GRU_J) ? 0 : J + 1;
            gru3_issue_slab<WAVES>(R2frag + (size_t)nJ * GRU2_SLAB, slab[buf ^ 1], wave, (unsigned)lane * 16u);

            // one base per gate: the ds_read offset field is 16 bits and the n-gate blocks start at 50 KB
            const unsigned rdz = slab_addr[buf], rdr = rdz + GRU_J * 1024, rdn = rdz + 2 * GRU_J * 1024;
            // ring of D super-steps: slot S % D holds the z, r, n fragment blocks of super-step S
            f32x4 wz[D], wr[D], wn[D];
            StaticFor<0, D>::run([&](auto sc) {
                constexpr int S = decltype(sc)::value;
                lds_read_b128<S * 1024>(wz[S], rdz);
                lds_read_b128<S * 1024>(wr[S], rdr);
                lds_read_b128<S * 1024>(wn[S], rdn);
            });

            f32x4 az = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 ar = az, an = az;
            auto super_step = [&](auto sc) {
                constexpr int S = decltype(sc)::value;
                constexpr int k = S % D;
                // younger reads allowed in flight: those of super-steps S+1 .. S+D-1 (fewer near the end)
                constexpr int ahead = (GRU_J - 1 - S < D - 1) ? GRU_J - 1 - S : D - 1;
                constexpr int younger = (3 * ahead < 13) ? 3 * ahead : 13; // lgkmcnt is a 4-bit field
                lds_wait<younger + 2>(wz[k]);
                const f32x4 hv = hreg[S];
                az = MFMA16(wz[k].x, hv.x, az);
                lds_wait<younger + 1>(wr[k]);
                ar = MFMA16(wr[k].x, hv.x, ar);
                lds_wait<younger>(wn[k]);
                an = MFMA16(wn[k].x, hv.x, an);
                az = MFMA16(wz[k].y, hv.y, az);
                ar = MFMA16(wr[k].y, hv.y, ar);
                an = MFMA16(wn[k].y, hv.y, an);
                az = MFMA16(wz[k].z, hv.z, az);
                ar = MFMA16(wr[k].z, hv.z, ar);
                an = MFMA16(wn[k].z, hv.z, an);
                az = MFMA16(wz[k].w, hv.w, az);
                ar = MFMA16(wr[k].w, hv.w, ar);
                an = MFMA16(wn[k].w, hv.w, an);
                if (S + D < GRU_J) {
                    lds_read_b128<(S + D) * 1024>(wz[k], rdz);
                    lds_read_b128<(S + D) * 1024>(wr[k], rdr);
                    lds_read_b128<(S + D) * 1024>(wn[k], rdn);
                }
            };
            // The gate operands are requested late in the tile (their registers are live only for the
            // last super-steps) but still several thousand cycles before the gate math needs them.
            constexpr int LOAD_AT = 18;
            StaticFor<0, LOAD_AT>::run(super_step);
            const f32x4 giz = ld4(gi_t + 192 * J, gi_off);
            const f32x4 gir = ld4(gi_t + 192 * J + 64, gi_off);
            const f32x4 gin = ld4(gi_t + 192 * J + 128, gi_off);
            const f32x4 hp = ld4(h_prev + 64 * J, h_off);
            const f32x4 bn = ld4(bR_b + 64 * J + 8 * GRU_H, b_off);
            StaticFor<LOAD_AT, GRU_J>::run(super_step);
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fast_sigmoid(giz[r] + az[r]);
                const float rr = fast_sigmoid(gir[r] + ar[r]);
                const float n = fast_tanh(gin[r] + rr * (an[r] + bn[r]));
                h[r] = (1.0f - z) * n + z * hp[r];
            }
            {
                unsigned o = h_off;
                asm volatile("" : "+v"(o));
                *(__attribute__((address_space(1))) f32x4*)(h_out + 64 * J + o) = h;
            }
            // next slab landed (everything older than the h store(s) has retired) and everyone is done
            // reading this one (all ds_reads were waited for above)
            if (TS3) {
                store_split3(hs_w + (size_t)t * (13 * 3072), J, h);
                asm volatile("s_waitcnt vmcnt(4)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(1)\n\ts_barrier" ::: "memory");
            }
            buf ^= 1;
        }
    }
}

// ------------------------------------------------------------------ GRU recurrence, low latency
// For small batches (a push of a few 0.5 s chunks) the recurrence is a latency problem: 53 dependent
// steps.  Here one workgroup owns 16 sequences and its 8 wavefronts split the 25 unit tiles of a
// step between them (wave w: tiles w, w+8, w+16, and 24 for wave 0), so all four SIMDs of the CU work
// on the same step: ~7 tiles x 300 MFMAs per SIMD per step instead of 25 tiles on one SIMD.
//   * nothing is shared between the waves except h, so weight fragments go straight from L2 into
//     VGPRs (coalesced 1 KB blocks of R2frag) through a ring of five super-steps that runs across tile
//     boundaries (25 % 5 == 0 keeps the slot of every (tile, super-step) static);
//   * h_t is exchanged through LDS in operand layout hs[S][lane] (the float4 a lane writes for unit
//     tile J is the float4 the same lane index reads as super-step S = J), double-buffered, one
//     barrier per step.
// gi holds Wx + Wb; Rb is added here.
// (the body is a device function: gru_lat_kernel runs it for one layer, gru_ws2_fallback_kernel for both)
// RT: row tiles (16 sequences each) per workgroup.  RT = 1: h_{t-1} of the step is copied into 100 registers up front.
// RT = 2, 3 (gru_lat2_kernel, gru_lat3_kernel: launches of more 16-sequence tiles than CUs): what paces this kernel is the 1.92 MB of R every
// workgroup re-streams from L2 per step, so the row tiles share ONE stream -- every fragment feeds all their chains -- and
// h_{t-1} is read from LDS where a super-step uses it (the registers hold RT sets of accumulators instead).  An output is
// the same chain of fmas either way: the same bits.
template <int WAVES, int RT = 1>
__device__ __forceinline__ void gru_lat_body(const float* gi, const float* __restrict__ R2frag, const float* __restrict__ bR,
                                             float* hout, int T, int gi_js, int gi_gs, float* hs /* [2][RT][GRU_J * 256] */)
{
    // gi_js / gi_gs: floats between unit tiles / between gates in a gi row (16, 400: gate-major rows of the
    // small-batch GEMM; 48, 16: the tile-major rows of the large-batch GEMM)
    typedef const __attribute__((address_space(1))) f32x4* gptr4;
    constexpr int D = 5;
    constexpr bool HREG = RT == 1;
    constexpr int HB = GRU_J * 256; // floats of one row tile's h in operand layout

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    const float* gi_seq[RT];
    float* h_seq[RT];
#pragma unroll
    for (int rt = 0; rt < RT; ++rt) {
        const size_t seq = ((size_t)blockIdx.x * RT + rt) * 16 + m;
        gi_seq[rt] = gi + seq * T * (3 * GRU_H) + 4 * q;
        h_seq[rt] = hout + seq * T * GRU_H + 4 * q;
    }
    const float* bR_q = bR + 4 * q;

    // ---- t = 0: h_{-1} = 0, so R h + Rb = Rb
    for (int J = wave; J < GRU_J; J += WAVES) {
        const f32x4 bz = *reinterpret_cast<const f32x4*>(bR_q + 16 * J);
        const f32x4 br = *reinterpret_cast<const f32x4*>(bR_q + GRU_H + 16 * J);
        const f32x4 bn = *reinterpret_cast<const f32x4*>(bR_q + 2 * GRU_H + 16 * J);
#pragma unroll
        for (int rt = 0; rt < RT; ++rt) {
            const f32x4 giz = *reinterpret_cast<const f32x4*>(gi_seq[rt] + gi_js * J);
            const f32x4 gir = *reinterpret_cast<const f32x4*>(gi_seq[rt] + gi_gs + gi_js * J);
            const f32x4 gin = *reinterpret_cast<const f32x4*>(gi_seq[rt] + 2 * gi_gs + gi_js * J);
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fast_sigmoid(giz[r] + bz[r]);
                const float rr = fast_sigmoid(gir[r] + br[r]);
                const float n = fast_tanh(gin[r] + rr * bn[r]);
                h[r] = (1.0f - z) * n + z * 0.0f;
            }
            *reinterpret_cast<f32x4*>(h_seq[rt] + 16 * J) = h;
            reinterpret_cast<f32x4*>(hs + rt * HB)[J * 64 + lane] = h;
        }
    }
    __syncthreads();
    int cur = 0;

    // fragment pointer of (tile J, gate g, super-step S): R2frag[J][g][S][lane][4]
    auto frag = [&](int J, int g, int S) -> gptr4 {
        return (gptr4)(R2frag + (size_t)J * GRU2_SLAB + (size_t)(g * GRU_J + S) * 256 + lane * 4);
    };

    // The fragment ring runs across unit tiles *and* across time steps (the weights do not depend on h):
    // after a wavefront's last tile of a step it is refilled with the first tile of the next step, so
    // a step starts with its first five super-steps already on the way.
    f32x4 wz[D], wr[D], wn[D];
#pragma unroll
    for (int S = 0; S < D; ++S) {
        wz[S] = *frag(wave, 0, S);
        wr[S] = *frag(wave, 1, S);
        wn[S] = *frag(wave, 2, S);
    }
    for (int t = 1; t < T; ++t) {
        const f32x4* hcur = reinterpret_cast<const f32x4*>(hs + cur * (RT * HB)) + lane;
        f32x4* hnxt = reinterpret_cast<f32x4*>(hs + (cur ^ 1) * (RT * HB)) + lane;

        f32x4 hreg[HREG ? GRU_J : 1];
        if (HREG) {
#pragma unroll
            for (int S = 0; S < GRU_J; ++S) hreg[S] = hcur[S * 64];
        }

        for (int J = wave; J < GRU_J; J += WAVES) {
            const int Jn = (J + WAVES < GRU_J) ? J + WAVES : wave; // this wave's next tile; after the last one, its first tile of the next step
            // two accumulation chains per gate (even / odd super-steps), summed at the end: the order of
            // gru_ws_kernel, whose fallback this kernel is -- both give the same bits
            f32x4 az[RT][2], ar[RT][2], an[RT][2];
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) az[rt][0] = az[rt][1] = ar[rt][0] = ar[rt][1] = an[rt][0] = an[rt][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            f32x4 giz[RT], gir[RT], gin[RT], hp[RT], bz, br, bn;
#pragma unroll
            for (int S = 0; S < GRU_J; ++S) {
                const int k = S % D;
                const int c = S & 1;
                f32x4 hv[RT];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) hv[rt] = HREG ? hreg[HREG ? S : 0] : hcur[(rt * GRU_J + S) * 64];
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    az[rt][c] = MFMA16(wz[k].x, hv[rt].x, az[rt][c]);
                    ar[rt][c] = MFMA16(wr[k].x, hv[rt].x, ar[rt][c]);
                    an[rt][c] = MFMA16(wn[k].x, hv[rt].x, an[rt][c]);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    az[rt][c] = MFMA16(wz[k].y, hv[rt].y, az[rt][c]);
                    ar[rt][c] = MFMA16(wr[k].y, hv[rt].y, ar[rt][c]);
                    an[rt][c] = MFMA16(wn[k].y, hv[rt].y, an[rt][c]);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    az[rt][c] = MFMA16(wz[k].z, hv[rt].z, az[rt][c]);
                    ar[rt][c] = MFMA16(wr[k].z, hv[rt].z, ar[rt][c]);
                    an[rt][c] = MFMA16(wn[k].z, hv[rt].z, an[rt][c]);
                }
#pragma unroll
                for (int rt = 0; rt < RT; ++rt) {
                    az[rt][c] = MFMA16(wz[k].w, hv[rt].w, az[rt][c]);
                    ar[rt][c] = MFMA16(wr[k].w, hv[rt].w, ar[rt][c]);
                    an[rt][c] = MFMA16(wn[k].w, hv[rt].w, an[rt][c]);
                }
                // refill the slot with the super-step D ahead; past the end of this tile that is the
                // next tile's super-step S + D - 25
                if (S + D < GRU_J) {
                    wz[k] = *frag(J, 0, S + D);
                    wr[k] = *frag(J, 1, S + D);
                    wn[k] = *frag(J, 2, S + D);
                } else {
                    wz[k] = *frag(Jn, 0, S + D - GRU_J);
                    wr[k] = *frag(Jn, 1, S + D - GRU_J);
                    wn[k] = *frag(Jn, 2, S + D - GRU_J);
                }
                if (S == 15) {
#pragma unroll
                    for (int rt = 0; rt < RT; ++rt) {
                        const float* gi_t = gi_seq[rt] + (size_t)t * (3 * GRU_H);
                        giz[rt] = *reinterpret_cast<const f32x4*>(gi_t + gi_js * J);
                        gir[rt] = *reinterpret_cast<const f32x4*>(gi_t + gi_gs + gi_js * J);
                        gin[rt] = *reinterpret_cast<const f32x4*>(gi_t + 2 * gi_gs + gi_js * J);
                        hp[rt] = hcur[(rt * GRU_J + J) * 64];
                    }
                    bz = *reinterpret_cast<const f32x4*>(bR_q + 16 * J);
                    br = *reinterpret_cast<const f32x4*>(bR_q + GRU_H + 16 * J);
                    bn = *reinterpret_cast<const f32x4*>(bR_q + 2 * GRU_H + 16 * J);
                }
            }
#pragma unroll
            for (int rt = 0; rt < RT; ++rt) {
                const f32x4 sz = az[rt][0] + az[rt][1], sr = ar[rt][0] + ar[rt][1], sn = an[rt][0] + an[rt][1];
                f32x4 h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float z = fast_sigmoid(giz[rt][r] + (sz[r] + bz[r]));
                    const float rr = fast_sigmoid(gir[rt][r] + (sr[r] + br[r]));
                    const float n = fast_tanh(gin[rt][r] + rr * (sn[r] + bn[r]));
                    h[r] = (1.0f - z) * n + z * hp[rt][r];
                }
                *reinterpret_cast<f32x4*>(h_seq[rt] + (size_t)t * GRU_H + 16 * J) = h;
                hnxt[(rt * GRU_J + J) * 64] = h;
            }
        }
        __syncthreads(); // h_t complete in the other buffer; everyone has read this one
        cur ^= 1;
    }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void gru_lat_kernel(const float* __restrict__ gi,
                                                             const float* __restrict__ R2frag,
                                                             const float* __restrict__ bR,
                                                             float* hout, int T, const unsigned* guard,
                                                             int gi_js, int gi_gs)
{
    __shared__ __attribute__((aligned(16))) float hs[2][GRU_J * 256];
    // launched behind gru_ws_kernel as its fallback: runs only if that kernel raised *guard (kernels_ws.hip)
    if (guard && *guard == 0) return;
    gru_lat_body<WAVES>(gi, R2frag, bR, hout, T, gi_js, gi_gs, &hs[0][0]);
}

// two row tiles per workgroup, one stream of R for both (gru_lat_body<8, 2>): 102 KB of dynamic LDS
__global__ __launch_bounds__(512) void gru_lat2_kernel(const float* __restrict__ gi, const float* __restrict__ R2frag,
                                                       const float* __restrict__ bR, float* hout, int T, int gi_js, int gi_gs)
{
    extern __shared__ __attribute__((aligned(16))) float lat2_hs[];
    gru_lat_body<8, 2>(gi, R2frag, bR, hout, T, gi_js, gi_gs, lat2_hs);
}
// three row tiles (154 KB of LDS, 190 VGPRs): 8193..12288 sequences in one round
__global__ __launch_bounds__(512) void gru_lat3_kernel(const float* __restrict__ gi, const float* __restrict__ R2frag,
                                                       const float* __restrict__ bR, float* hout, int T, int gi_js, int gi_gs)
{
    extern __shared__ __attribute__((aligned(16))) float lat3_hs[];
    gru_lat_body<8, 3>(gi, R2frag, bR, hout, T, gi_js, gi_gs, lat3_hs);
}

// The whole fallback of the pipelined two-layer recurrence (kernels_ws.hip: gru_ws2_kernel / gru_ws2k_kernel) in ONE
// launch behind it -- it used to be three guarded launches (gru_lat, layer 2's input projection, gru_lat) and a fourth
// that counted the pass, each ~4.5 us of a 0.5 ms call even when it returns at once.  A workgroup owns 16 sequences
// through both layers: layer 1's recurrence (gru_lat_body), then gi2 = W_ih h1 + Wb for its own rows -- the
// instruction sequence of panel_gemm_s_kernel per output: the k-ordered chain from zero, the bias added after it, so
// the bits are the three-launch chain's -- written over its gi rows, then layer 2's recurrence.  Nothing crosses
// workgroups.  It runs only if the error word was raised.  Either way every workgroup takes a ticket when it is done
// (after it has read the error word), and the last one adds the pass to the fallback counter and zeroes the polled
// words for the next pass, so the sequence needs no separate reset launch in front and no count launch behind.
//   sync: [0, 512) the two layers' flags, [512] the error word, [513] the ticket counter.
__global__ __launch_bounds__(512) void gru_ws2_fallback_kernel(float* gi, const float* __restrict__ feat, const float* __restrict__ W1frag_nt2,
                                                               const float* __restrict__ bG1, const float* __restrict__ R1frag, const float* __restrict__ bR1,
                                                               const float* __restrict__ W2frag_nt2, const float* __restrict__ bW2,
                                                               const float* __restrict__ R2frag, const float* __restrict__ bR2,
                                                               float* h1, float* h2, int T, unsigned* sync,
                                                               unsigned long long* fallbacks, int zero_at, int zero_n)
{
    __shared__ __attribute__((aligned(16))) float hs[2][GRU_J * 256];
    __shared__ int s_last;
    const bool run = *(const volatile unsigned*)(sync + 512) != 0u; // uniform: nobody writes the word while this launch runs
    if (run) {
        const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), m = lane & 15, q = lane >> 4;
        const size_t row0 = ((size_t)blockIdx.x * 16 + m) * T;
        if (feat) {
            // the pipelined kernel computed layer 1's input projection itself (gru_ws2k_kernel): gi1 = W' x + b' for this
            // workgroup's rows first, the instruction sequence of the GEMM that otherwise runs in front (K = 176: 11 super-steps)
            constexpr int XS = kFeatStride / 16;
            for (int t = 0; t < T; ++t) {
                const float* a_ptr = feat + (row0 + t) * kFeatStride + 4 * q;
                float* c_ptr = gi + (row0 + t) * (3 * GRU_H) + 4 * q;
                for (int u = wave; u < 3 * GRU_J; u += 8) {
                    const f32x4* wf = reinterpret_cast<const f32x4*>(W1frag_nt2) + ((size_t)(u >> 1) * XS * 2 + (u & 1)) * 64 + lane;
                    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int s = 0; s < XS; ++s) {
                        const f32x4 w4 = wf[(size_t)s * 2 * 64];
                        const f32x4 a = *reinterpret_cast<const f32x4*>(a_ptr + 16 * s);
                        acc = MFMA16(w4.x, a.x, acc);
                        acc = MFMA16(w4.y, a.y, acc);
                        acc = MFMA16(w4.z, a.z, acc);
                        acc = MFMA16(w4.w, a.w, acc);
                    }
                    *reinterpret_cast<f32x4*>(c_ptr + 16 * u) = acc + *reinterpret_cast<const f32x4*>(bG1 + 16 * u + 4 * q);
                }
            }
            __threadfence();
            __syncthreads();
        }
        gru_lat_body<8>(gi, R1frag, bR1, h1, T, 48, 16, &hs[0][0]);
        __threadfence(); // h1 rows written by the other wavefronts of this workgroup
        __syncthreads();
        for (int t = 0; t < T; ++t) {
            const float* a_ptr = h1 + (row0 + t) * GRU_H + 4 * q;
            float* c_ptr = gi + (row0 + t) * (3 * GRU_H) + 4 * q;
            for (int u = wave; u < 3 * GRU_J; u += 8) { // unit tiles of the 1200 outputs (tile-major order: in the weights' row order)
                const f32x4* wf = reinterpret_cast<const f32x4*>(W2frag_nt2) + ((size_t)(u >> 1) * GRU_J * 2 + (u & 1)) * 64 + lane;
                f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 5
                for (int s = 0; s < GRU_J; ++s) {
                    const f32x4 w4 = wf[(size_t)s * 2 * 64];
                    const f32x4 a = *reinterpret_cast<const f32x4*>(a_ptr + 16 * s);
                    acc = MFMA16(w4.x, a.x, acc);
                    acc = MFMA16(w4.y, a.y, acc);
                    acc = MFMA16(w4.z, a.z, acc);
                    acc = MFMA16(w4.w, a.w, acc);
                }
                *reinterpret_cast<f32x4*>(c_ptr + 16 * u) = acc + *reinterpret_cast<const f32x4*>(bW2 + 16 * u + 4 * q);
            }
        }
        __threadfence(); // gi rows rewritten: this CU's L1 may still hold layer 1's gi in those lines
        __syncthreads();
        gru_lat_body<8>(gi, R2frag, bR2, h2, T, 48, 16, &hs[0][0]);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        s_last = atomicAdd(sync + 513, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (s_last) {
        if (threadIdx.x == 0 && run) *fallbacks += 1ull;
        for (int i = threadIdx.x; i < 514; i += 512) sync[i] = 0u;
        for (int i = threadIdx.x; i < zero_n; i += 512) sync[zero_at + i] = 0u; // gru_ws2k's XCD-local flags and placement tickets
    }
}

int fvad_launch_gru_ws2_fallback(float* gi, const float* feat, const float* W1frag_nt2, const float* bG1, const float* R1frag, const float* bR1,
                                 const float* W2frag_nt2, const float* bW2, const float* R2frag, const float* bR2, float* h1, float* h2,
                                 long n_seq_pad, int T, unsigned* sync, unsigned long long* fallbacks, hipStream_t stream, int zero_at, int zero_n)
{
    if (n_seq_pad <= 0 || n_seq_pad % 16) return -1;
    hipLaunchKernelGGL(gru_ws2_fallback_kernel, dim3((unsigned)(n_seq_pad / 16)), dim3(512), 0, stream, gi, feat, W1frag_nt2, bG1, R1frag, bR1, W2frag_nt2, bW2,
                       R2frag, bR2, h1, h2, T, sync, fallbacks, zero_at, zero_n);
    return 0;
}

int fvad_launch_gru_lat(const float* gi, const float* R2frag, const float* bR, float* hout,
                        long n_seq_pad, int T, const unsigned* guard, int tile_major, hipStream_t stream, int row_tiles)
{
    if (n_seq_pad % 16) return -1;
    if (row_tiles == 3 && !guard && n_seq_pad % 48 == 0) {
        constexpr size_t lds3 = (size_t)2 * 3 * GRU_J * 256 * sizeof(float);
        if (hipFuncSetAttribute((const void*)gru_lat3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds3) != hipSuccess) return -2;
        hipLaunchKernelGGL(gru_lat3_kernel, dim3((unsigned)(n_seq_pad / 48)), dim3(512), lds3, stream, gi, R2frag, bR, hout, T,
                           tile_major ? 48 : 16, tile_major ? 16 : GRU_H);
        return 0;
    }
    if (row_tiles == 2 && !guard && n_seq_pad % 32 == 0) { // same bits; for launches of more 16-sequence tiles than CUs
        constexpr size_t lds = (size_t)2 * 2 * GRU_J * 256 * sizeof(float);
        if (hipFuncSetAttribute((const void*)gru_lat2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        hipLaunchKernelGGL(gru_lat2_kernel, dim3((unsigned)(n_seq_pad / 32)), dim3(512), lds, stream, gi, R2frag, bR, hout, T,
                           tile_major ? 48 : 16, tile_major ? 16 : GRU_H);
        return 0;
    }
    hipLaunchKernelGGL((gru_lat_kernel<8>), dim3((unsigned)(n_seq_pad / 16)), dim3(512), 0, stream, gi, R2frag, bR, hout, T, guard,
                       tile_major ? 48 : 16, tile_major ? 16 : GRU_H);
    return 0;
}

// ------------------------------------------------------------------ GRU recurrence, any hidden size
// NSNet2.init binds whatever ONNX file the configuration names (src/NSNet2.zig:53-112, VADPipeline.zig:25); the
// kernels above are specialised for the baseline's H = 400.  This one takes the hidden size from the file: H padded
// to J unit tiles of 16 (padded units have zero weights and biases: z = 1/2, n = 0, so they stay 0 from h_0 = 0 on).
// gru_lat_kernel's scheme with run-time loops: one workgroup = 16 sequences, its 8 wavefronts share the J unit tiles
// of a step, h_{t-1} sits in LDS in operand layout (double-buffered, one barrier per step), weight fragments come
// straight from L2 one super-step ahead, two accumulation chains per gate.
//   gi   [n_seq_pad * T][gi_ld]: gate g of unit tile j at g * 16 J + 16 j (Wx + Wb);  bR [3][16 J] (Rb, padded)
//   R2frag [J][3 g][J S][64][4] (pack_gru_r2 of the padded matrix);  hout [n_seq_pad * T][h_ld]
__global__ __launch_bounds__(512) void gru_gen_kernel(const float* __restrict__ gi, int gi_ld,
                                                      const float* __restrict__ R2frag, const float* __restrict__ bR,
                                                      float* hout, int h_ld, int T, int J)
{
    extern __shared__ __attribute__((aligned(16))) float hs_dyn[]; // [2][J][64] float4
    typedef const __attribute__((address_space(1))) f32x4* gptr4;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = lane & 15;
    const int q = lane >> 4;
    const int Hp = 16 * J;
    const size_t seq = (size_t)blockIdx.x * 16 + m;
    const float* gi_seq = gi + seq * T * (size_t)gi_ld + 4 * q;
    float* h_seq = hout + seq * T * (size_t)h_ld + 4 * q;
    const float* bR_q = bR + 4 * q;
    f32x4* hs = reinterpret_cast<f32x4*>(hs_dyn);
    const f32x4 zero4 = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int j = wave; j < J; j += 8) { // t = 0: h_{-1} = 0, so R h + Rb = Rb
        const f32x4 giz = *reinterpret_cast<const f32x4*>(gi_seq + 16 * j);
        const f32x4 gir = *reinterpret_cast<const f32x4*>(gi_seq + Hp + 16 * j);
        const f32x4 gin = *reinterpret_cast<const f32x4*>(gi_seq + 2 * Hp + 16 * j);
        const f32x4 bz = *reinterpret_cast<const f32x4*>(bR_q + 16 * j);
        const f32x4 br = *reinterpret_cast<const f32x4*>(bR_q + Hp + 16 * j);
        const f32x4 bn = *reinterpret_cast<const f32x4*>(bR_q + 2 * Hp + 16 * j);
        f32x4 h;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float z = fast_sigmoid(giz[r] + bz[r]);
            const float rr = fast_sigmoid(gir[r] + br[r]);
            const float n = fast_tanh(gin[r] + rr * bn[r]);
            h[r] = (1.0f - z) * n + z * 0.0f;
        }
        *reinterpret_cast<f32x4*>(h_seq + 16 * j) = h;
        hs[j * 64 + lane] = h;
    }
    __syncthreads();
    int cur = 0;
    for (int t = 1; t < T; ++t) {
        const float* gi_t = gi_seq + (size_t)t * gi_ld;
        float* h_out = h_seq + (size_t)t * h_ld;
        const f32x4* hcur = hs + cur * (J * 64) + lane;
        f32x4* hnxt = hs + (cur ^ 1) * (J * 64) + lane;
        for (int j = wave; j < J; j += 8) {
            gptr4 fz = (gptr4)(R2frag + ((size_t)(j * 3 + 0) * J) * 256 + lane * 4);
            gptr4 fr = (gptr4)(R2frag + ((size_t)(j * 3 + 1) * J) * 256 + lane * 4);
            gptr4 fn = (gptr4)(R2frag + ((size_t)(j * 3 + 2) * J) * 256 + lane * 4);
            const f32x4 giz = *reinterpret_cast<const f32x4*>(gi_t + 16 * j);
            const f32x4 gir = *reinterpret_cast<const f32x4*>(gi_t + Hp + 16 * j);
            const f32x4 gin = *reinterpret_cast<const f32x4*>(gi_t + 2 * Hp + 16 * j);
            f32x4 az[2] = {zero4, zero4}, ar[2] = {zero4, zero4}, an[2] = {zero4, zero4};
            f32x4 wz = fz[0], wr = fr[0], wn = fn[0];
            for (int S = 0; S < J; ++S) {
                const int Sn = (S + 1 < J) ? S + 1 : S;
                const f32x4 wz1 = fz[Sn * 64], wr1 = fr[Sn * 64], wn1 = fn[Sn * 64]; // one super-step ahead
                const f32x4 hv = hcur[S * 64];
                const int c = S & 1;
                az[c] = MFMA16(wz.x, hv.x, az[c]);
                ar[c] = MFMA16(wr.x, hv.x, ar[c]);
                an[c] = MFMA16(wn.x, hv.x, an[c]);
                az[c] = MFMA16(wz.y, hv.y, az[c]);
                ar[c] = MFMA16(wr.y, hv.y, ar[c]);
                an[c] = MFMA16(wn.y, hv.y, an[c]);
                az[c] = MFMA16(wz.z, hv.z, az[c]);
                ar[c] = MFMA16(wr.z, hv.z, ar[c]);
                an[c] = MFMA16(wn.z, hv.z, an[c]);
                az[c] = MFMA16(wz.w, hv.w, az[c]);
                ar[c] = MFMA16(wr.w, hv.w, ar[c]);
                an[c] = MFMA16(wn.w, hv.w, an[c]);
                wz = wz1; wr = wr1; wn = wn1;
            }
            const f32x4 bz = *reinterpret_cast<const f32x4*>(bR_q + 16 * j);
            const f32x4 br = *reinterpret_cast<const f32x4*>(bR_q + Hp + 16 * j);
            const f32x4 bn = *reinterpret_cast<const f32x4*>(bR_q + 2 * Hp + 16 * j);
            const f32x4 hp = hcur[j * 64];
            const f32x4 sz = az[0] + az[1], sr = ar[0] + ar[1], sn = an[0] + an[1];
            f32x4 h;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float z = fast_sigmoid(giz[r] + (sz[r] + bz[r]));
                const float rr = fast_sigmoid(gir[r] + (sr[r] + br[r]));
                const float n = fast_tanh(gin[r] + rr * (sn[r] + bn[r]));
                h[r] = (1.0f - z) * n + z * hp[r];
            }
            *reinterpret_cast<f32x4*>(h_out + 16 * j) = h;
            hnxt[j * 64] = h;
        }
        __syncthreads(); // h_t complete in the other buffer; everyone has read this one
        cur ^= 1;
    }
}

int fvad_launch_gru_gen(const float* gi, int gi_ld, const float* R2frag, const float* bR, float* hout, int h_ld,
                        long n_seq_pad, int T, int J, hipStream_t stream)
{
    if (n_seq_pad % 16 || J < 1 || J > 64) return -1;
    const size_t lds = (size_t)2 * J * 1024;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)gru_gen_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(gru_gen_kernel, dim3((unsigned)(n_seq_pad / 16)), dim3(512), lds, stream, gi, gi_ld, R2frag, bR, hout, h_ld, T, J);
    return 0;
}

// hs3 != nullptr: h is also written as three-piece bf16 fragments (TS3, kernels_b3.hip)
int fvad_launch_gru_rec3(const float* gi, const float* R2frag, const float* bR, float* hout,
                         long n_seq_pad, int T, int waves, hipStream_t stream, float* hs3)
{
#define REC3(W_)                                                                                                          \
    if (waves == W_) {                                                                                                    \
        if (hs3) hipLaunchKernelGGL((gru_rec3_kernel<W_, 2, true>), dim3((unsigned)(n_seq_pad / (16 * W_))), dim3(64 * W_), 0, stream, gi, R2frag, bR, hout, T, hs3); \
        else hipLaunchKernelGGL((gru_rec3_kernel<W_, 2, false>), dim3((unsigned)(n_seq_pad / (16 * W_))), dim3(64 * W_), 0, stream, gi, R2frag, bR, hout, T, hs3); \
        return 0;                                                                                                         \
    }
    REC3(12)
    REC3(8)
    REC3(4)
#undef REC3
    return -1;
}

