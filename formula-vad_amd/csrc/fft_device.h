// fft_device.h -- device-side building blocks shared by the spectral translation units (kernels_stft.hip: K1 / K3,
// kernels_vadfft.hip: K4, kernels_fft.hip: the batch transforms, kernels_fftgen.hip: any even size).
//
// All wavefront FFTs share one scheme.  A real transform of length 2N is one complex
// transform of length N = R * L over z[n] = x[2n] + i x[2n+1] plus the same un-mixing pass
// kissfft uses ("super twiddles").  The complex transform keeps R points per lane in registers
// and spreads L points over lanes (N = 160: R = 5, L = 32, two frames per 64-lane wavefront;
// N = 512: R = 8, L = 64):
//     X[k1 + R k2] = sum_p W_L^{p k2} ( W_N^{p k1} sum_j z[p + L j] W_R^{j k1} )
// i.e. an R-point DFT in registers, one twiddle multiply, then R independent L-point
// decimation-in-frequency FFTs whose butterflies are lane exchanges.  Twiddles are
// read once per wavefront from tables the host evaluated in double (as kissfft does) and kept in
// registers.  f32 throughout; fp contraction is off for these files so products and sums round
// exactly where the reference's do.  (The 16- and 8-lane layouts of vadfft1024_band_kernel and
// rfft320_batch8_kernel put the register transform LAST or FIRST respectively and are described there.)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
#include "device_math.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2v __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct cpx { float r, i; };

__device__ __forceinline__ cpx cmul(cpx a, cpx b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__device__ __forceinline__ cpx cadd(cpx a, cpx b) { return {a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ cpx csub(cpx a, cpx b) { return {a.r - b.r, a.i - b.i}; }
__device__ __forceinline__ cpx cconj(cpx a) { return {a.r, -a.i}; }
template <bool INV> __device__ __forceinline__ cpx mul_mi(cpx a) // a * (-i) forward, a * (+i) inverse
{
    return INV ? cpx{-a.i, a.r} : cpx{a.i, -a.r};
}
__device__ __forceinline__ cpx ld_tw(const float* t, int idx) { return {t[2 * idx], t[2 * idx + 1]}; }
// twiddle multiply inside the wavefront FFT: one rounding less per component than cmul (fused multiply-add)
__device__ __forceinline__ cpx cmul_fma(cpx a, cpx b)
{
    return {__builtin_fmaf(a.r, b.r, -(a.i * b.i)), __builtin_fmaf(a.r, b.i, a.i * b.r)};
}

// ---- R-point DFT in registers (exponent sign: -, or + when INV)
template <bool INV> __device__ __forceinline__ void dft5(cpx (&v)[5])
{
    // W5^1 = (c1, -+s1), W5^2 = (c2, -+s2); constants rounded from double
    const float c1 = 0.30901699437494742f, s1 = 0.95105651629515357f;
    const float c2 = -0.80901699437494742f, s2 = 0.58778525229247313f;
    const float ya_i = INV ? s1 : -s1, yb_i = INV ? s2 : -s2;
    const cpx z0 = v[0];
    const cpx s7 = cadd(v[1], v[4]), s10 = csub(v[1], v[4]);
    const cpx s8 = cadd(v[2], v[3]), s9 = csub(v[2], v[3]);
    v[0] = {z0.r + (s7.r + s8.r), z0.i + (s7.i + s8.i)};
    const cpx s5 = {z0.r + s7.r * c1 + s8.r * c2, z0.i + s7.i * c1 + s8.i * c2};
    const cpx s6 = {s10.i * ya_i + s9.i * yb_i, -(s10.r * ya_i) - s9.r * yb_i};
    v[1] = csub(s5, s6);
    v[4] = cadd(s5, s6);
    const cpx s11 = {z0.r + s7.r * c2 + s8.r * c1, z0.i + s7.i * c2 + s8.i * c1};
    const cpx s12 = {-(s10.i * yb_i) + s9.i * ya_i, s10.r * yb_i - s9.r * ya_i};
    v[2] = cadd(s11, s12);
    v[3] = csub(s11, s12);
}

template <bool INV> __device__ __forceinline__ void dft4(cpx& c0, cpx& c1, cpx& c2, cpx& c3)
{
    const cpx e0 = cadd(c0, c2), e1 = csub(c0, c2);
    const cpx o0 = cadd(c1, c3), o1 = mul_mi<INV>(csub(c1, c3));
    c0 = cadd(e0, o0);
    c2 = csub(e0, o0);
    c1 = cadd(e1, o1);
    c3 = csub(e1, o1);
}

template <bool INV> __device__ __forceinline__ void dft8(cpx (&v)[8])
{
    const float h = 0.70710678118654752f;
    cpx a0 = cadd(v[0], v[4]), a1 = cadd(v[1], v[5]), a2 = cadd(v[2], v[6]), a3 = cadd(v[3], v[7]);
    cpx b0 = csub(v[0], v[4]), b1 = csub(v[1], v[5]), b2 = csub(v[2], v[6]), b3 = csub(v[3], v[7]);
    // b_j *= W8^j
    const cpx w1 = INV ? cpx{h, h} : cpx{h, -h};
    const cpx w3 = INV ? cpx{-h, h} : cpx{-h, -h};
    b1 = cmul(b1, w1);
    b2 = mul_mi<INV>(b2);
    b3 = cmul(b3, w3);
    dft4<INV>(a0, a1, a2, a3); // Y[0], Y[2], Y[4], Y[6]
    dft4<INV>(b0, b1, b2, b3); // Y[1], Y[3], Y[5], Y[7]
    v[0] = a0; v[2] = a1; v[4] = a2; v[6] = a3;
    v[1] = b0; v[3] = b1; v[5] = b2; v[7] = b3;
}

// 16 points as 4 x 4: X[k1 + 4 k2] = sum_n2 W4^{n2 k2} ( W16^{n2 k1} sum_n1 x[4 n1 + n2] W4^{n1 k1} )
template <bool INV> __device__ __forceinline__ void dft16(cpx (&v)[16])
{
    // W16^m = (cos, -+sin)(2 pi m / 16), constants rounded from double
    const float c1 = 0.92387953251128674f, s1 = 0.38268343236508977f, h = 0.70710678118654752f;
    auto w16 = [&](int m) -> cpx {
        cpx w;
        switch (m) {
        case 1: w = {c1, -s1}; break;
        case 2: w = {h, -h}; break;
        case 3: w = {s1, -c1}; break;
        case 4: w = {0.0f, -1.0f}; break;
        case 6: w = {-h, -h}; break;
        default: w = {-c1, s1}; break; // m = 9
        }
        return INV ? cconj(w) : w;
    };
    cpx t[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        cpx a = v[n2], b = v[4 + n2], c = v[8 + n2], d = v[12 + n2];
        dft4<INV>(a, b, c, d);
        t[n2][0] = a; t[n2][1] = b; t[n2][2] = c; t[n2][3] = d;
    }
#pragma unroll
    for (int n2 = 1; n2 < 4; ++n2)
#pragma unroll
        for (int k1 = 1; k1 < 4; ++k1) t[n2][k1] = (n2 * k1 == 4) ? mul_mi<INV>(t[n2][k1]) : cmul(t[n2][k1], w16(n2 * k1));
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        cpx a = t[0][k1], b = t[1][k1], c = t[2][k1], d = t[3][k1];
        dft4<INV>(a, b, c, d);
        v[k1] = a; v[k1 + 4] = b; v[k1 + 8] = c; v[k1 + 12] = d;
    }
}

template <int R, bool INV> __device__ __forceinline__ void reg_dft(cpx (&v)[R])
{
    if constexpr (R == 5) dft5<INV>(v);
    else if constexpr (R == 4) dft4<INV>(v[0], v[1], v[2], v[3]);
    else if constexpr (R == 16) dft16<INV>(v);
    else dft8<INV>(v);
}

// Per-lane twiddle set of one wavefront FFT, loaded once and reused for every frame.
template <int R, int L> struct LaneTw {
    static constexpr int LOG_L = (L == 64) ? 6 : 5;
    cpx lane[R - 1];   // W_N^{p k1}, k1 = 1..R-1
    cpx stage[LOG_L];  // DIF stage twiddle of this lane: 1 for the lower half of a butterfly; for the upper half the
                       // twiddle (strides 16, 32) or MINUS the twiddle (strides 2..8, where wave_fft forms
                       // mine - other); unused for the last stage
};

template <int R, int L, bool INV>
__device__ __forceinline__ void lane_tw_load(LaneTw<R, L>& tw, const float* table /*[R*L][2] fwd*/, int p)
{
    constexpr int N = R * L;
#pragma unroll
    for (int k1 = 1; k1 < R; ++k1) {
        cpx t = ld_tw(table, p * k1);
        tw.lane[k1 - 1] = INV ? cconj(t) : t;
    }
    int s = 0;
#pragma unroll
    for (int h = L / 2; h >= 1; h >>= 1, ++s) {
        cpx t = {1.0f, 0.0f};
        if (p & h) {
            t = ld_tw(table, (p & (h - 1)) * (N / (2 * h)));
            if (INV) t = cconj(t);
            if (h < 16) t = {-t.r, -t.i}; // the DPP stages form mine - other on the upper lane (see wave_fft)
        }
        tw.stage[s] = t;
    }
}

// value of lane (l ^ H).  The FFT kernels are VALU-issue-bound (a wave64 VALU instruction holds its SIMD for
// four cycles, and ~80 % of all SIMD cycles of the batch FFT are VALU), while the LDS pipe is nearly idle.  The
// in-row strides therefore go through the LDS crossbar: ds_swizzle_b32 in bit-mask mode (lane ^ H inside groups
// of 32, no LDS memory, no address VGPR), one LDS-pipe instruction per exchange and NO VALU instruction -- the DPP
// forms (quad_perm / row_ror as v_mov_b32_dpp, two masked row shifts for H = 4) cost one to two VALU slots each.
// Strides 16 and 32 use gfx950's v_permlane16/32_swap on scalar pairs (swap_butterfly).
template <int H> __device__ __forceinline__ float lane_xor(float v, int lane)
{
    const int x = __float_as_int(v);
    int r;
    if constexpr (H < 16) {
        r = __builtin_amdgcn_ds_swizzle(x, (H << 10) | 0x1F); // and_mask 0x1f, or_mask 0, xor_mask H
    } else if constexpr (H == 16) {
        const auto sw = __builtin_amdgcn_permlane16_swap((unsigned)x, (unsigned)x, false, false);
        r = (int)((lane & 16) ? sw[0] : sw[1]);
    } else {
        static_assert(H == 32, "lane_xor: stride");
        const auto sw = __builtin_amdgcn_permlane32_swap((unsigned)x, (unsigned)x, false, false);
        r = (int)((lane & 32) ? sw[0] : sw[1]);
    }
    return __int_as_float(r);
}
__device__ __forceinline__ float lane_xor_dyn(float v, int h, int lane) // h is a constant after unrolling
{
    switch (h) {
    case 1: return lane_xor<1>(v, lane);
    case 2: return lane_xor<2>(v, lane);
    case 4: return lane_xor<4>(v, lane);
    case 8: return lane_xor<8>(v, lane);
    case 16: return lane_xor<16>(v, lane);
    default: return lane_xor<32>(v, lane);
    }
}

// DIF butterflies of TWO scalars x, y across lanes l and l ^ H for the row- and half-crossing strides, with
// gfx950's v_permlane16/32_swap (exchanges the odd rows / upper half of its first operand with the even rows /
// lower half of its second).  swap(x, y) leaves a = [x_low, y_low], b = [x_high, y_high] (position by lane half);
// a + b and a - b are then x's two results on the lower lanes and y's two on the upper lanes, and a second swap
// puts each result on the lane that owns it: 4 instructions for 2 scalars instead of 2 x (2 copies, swap,
// select, fma).  Values: lower lane x_low + x_high, upper lane x_low - x_high, exactly as before.
template <int H> __device__ __forceinline__ void swap_butterfly(float& x, float& y)
{
    static_assert(H == 16 || H == 32, "swap_butterfly: stride");
    const unsigned xi = (unsigned)__float_as_int(x), yi = (unsigned)__float_as_int(y);
    const auto sw = (H == 16) ? __builtin_amdgcn_permlane16_swap(xi, yi, false, false)
                              : __builtin_amdgcn_permlane32_swap(xi, yi, false, false);
    const float a = __int_as_float((int)sw[0]), b = __int_as_float((int)sw[1]);
    const float sum = a + b, dif = a - b;
    const unsigned si = (unsigned)__float_as_int(sum), di = (unsigned)__float_as_int(dif);
    const auto sw2 = (H == 16) ? __builtin_amdgcn_permlane16_swap(si, di, false, false)
                               : __builtin_amdgcn_permlane32_swap(si, di, false, false);
    x = __int_as_float((int)sw2[0]);
    y = __int_as_float((int)sw2[1]);
}

// v[j] = z[p + L j] in, v[k1] = Z[k1 + R * bitrev_L(p)] out.
template <int R, int L, bool INV>
__device__ __forceinline__ void wave_fft(cpx (&v)[R], const LaneTw<R, L>& tw, int p)
{
    reg_dft<R, INV>(v);
#pragma unroll
    for (int k1 = 1; k1 < R; ++k1) v[k1] = cmul_fma(v[k1], tw.lane[k1 - 1]);
    int s = 0;
#pragma unroll
    for (int h = L / 2; h >= 1; h >>= 1, ++s) {
        // DIF butterfly across lanes l and l ^ h: lower lane a + b, upper lane (a_low - a_high) * w.
        // With sgn = +1 on the lower and -1 on the upper lane both are one fma per component, exactly the
        // sum / difference (a product by +-1 is exact):
        //   h > 1:  t = other * sgn + mine  (upper: mine - other, the twiddle table holds -w there), written so that
        //           the lane exchange folds into the fma as a DPP operand;
        //   h = 1:  t = sgn * mine + other  (upper: other - mine; the last stage's twiddle is W^0 = 1).
        if (h >= 16) {
#pragma unroll
            for (int k1 = 0; k1 < R; ++k1) {
                if (h == 16) swap_butterfly<16>(v[k1].r, v[k1].i);
                else swap_butterfly<32>(v[k1].r, v[k1].i);
                v[k1] = cmul_fma(v[k1], tw.stage[s]);
            }
            continue;
        }
        const float sgn = (p & h) ? -1.0f : 1.0f;
#pragma unroll
        for (int k1 = 0; k1 < R; ++k1) {
            const cpx mine = v[k1];
            cpx other;
            other.r = lane_xor_dyn(mine.r, h, p); // bit h of p is bit h of the lane index for every h < L
            other.i = lane_xor_dyn(mine.i, h, p);
            if (h > 1) {
                const cpx t = {__builtin_fmaf(other.r, sgn, mine.r), __builtin_fmaf(other.i, sgn, mine.i)};
                v[k1] = cmul_fma(t, tw.stage[s]);
            } else {
                v[k1] = {__builtin_fmaf(sgn, mine.r, other.r), __builtin_fmaf(sgn, mine.i, other.i)};
            }
        }
    }
}

template <int L> __device__ __forceinline__ int bitrev_lane(int p)
{
    return (int)(__brev((unsigned)p) >> (L == 64 ? 26 : 27));
}

// kissfft's real-FFT un-mixing for one k in [1, ncfft/2]: writes X[k] and X[ncfft-k]
__device__ __forceinline__ void unmix_fwd(cpx zk, cpx znk, cpx st, cpx& xk, cpx& xnk)
{
    const cpx fpnk = cconj(znk);
    const cpx f1k = cadd(zk, fpnk);
    const cpx f2k = csub(zk, fpnk);
    const cpx tw = cmul(f2k, st);
    xk = {(f1k.r + tw.r) * 0.5f, (f1k.i + tw.i) * 0.5f};
    xnk = {(f1k.r - tw.r) * 0.5f, (tw.i - f1k.i) * 0.5f};
}
// The same un-mixing with the factor 1/2 folded into the table (sth = st / 2, exact) and the final sums as fmas:
// 0.5 * f1k + f2k * sth is (f1k + f2k * st) / 2 with the same roundings (products and their difference are halved
// exactly, and round(a / 2 + b / 2) = round(a + b) / 2), in 14 instructions instead of 18.  Entry 0 of the table,
// (0, -1/2), extends it to k = 0 with znk := z[0]: it yields X[0] = z.r + z.i and X[160] = z.r - z.i, kissfft's
// special case, without a branch.
__device__ __forceinline__ void unmix_fwd_h(cpx zk, cpx znk, cpx sth, cpx& xk, cpx& xnk)
{
    const cpx f1k = {zk.r + znk.r, zk.i - znk.i};
    const cpx f2k = {zk.r - znk.r, zk.i + znk.i};
    const cpx twh = cmul(f2k, sth);
    xk = {__builtin_fmaf(0.5f, f1k.r, twh.r), __builtin_fmaf(0.5f, f1k.i, twh.i)};
    xnk = {__builtin_fmaf(0.5f, f1k.r, -twh.r), __builtin_fmaf(-0.5f, f1k.i, twh.i)};
}
// and the inverse pre-mixing: T[k], T[ncfft-k] from Y[k], Y[ncfft-k]; st is the INVERSE twiddle
__device__ __forceinline__ void premix_inv(cpx fk, cpx fnk, cpx st_inv, cpx& tk, cpx& tnk)
{
    const cpx fnkc = cconj(fnk);
    const cpx fek = cadd(fk, fnkc);
    const cpx tmp = csub(fk, fnkc);
    const cpx fok = cmul(tmp, st_inv);
    tk = cadd(fek, fok);
    tnk = cconj(csub(fek, fok));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------- rfft-N, N = 128 R (K4's full-spectrum kernel, the N-point batch)
// One wavefront per frame of N = 512 / 1024 / 2048 samples (VADPipeline.Config.fft_size, VADPipeline.zig:21; the
// reference default is 1024): complex transform of length N/2 = R x 64 lanes, R = 4 / 8 / 16.
// mode: band sum only, or full N/2 + 1 magnitudes / bins too.
template <int R>
__device__ __forceinline__ void rfftN_wave(const float* __restrict__ x, const float* __restrict__ win,
                                           const VadFftPlan& pl, float* zl /*LDS [128 R]*/, int lane)
{
    LaneTw<R, 64> tw;
    lane_tw_load<R, 64, false>(tw, pl.tw, lane);
    cpx v[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int n = 2 * (lane + 64 * j);
        const float2 xv = *reinterpret_cast<const float2*>(x + n);
        const float2 wv = *reinterpret_cast<const float2*>(win + n);
        v[j] = {xv.x * wv.x, xv.y * wv.y};
    }
    wave_fft<R, 64, false>(v, tw, lane);
    const int k2 = bitrev_lane<64>(lane);
#pragma unroll
    for (int k1 = 0; k1 < R; ++k1) {
        const int k = k1 + R * k2;
        zl[2 * k] = v[k1].r;
        zl[2 * k + 1] = v[k1].i;
    }
}

// X[k], 0 <= k <= N/2, from the complex transform (length NC = N/2) in LDS
template <int R>
__device__ __forceinline__ cpx rfftN_bin(const float* zl, const float* st, int k)
{
    constexpr int NC = 64 * R;
    if (k == 0) return {zl[0] + zl[1], 0.0f};
    if (k == NC) return {zl[0] - zl[1], 0.0f};
    const int kk = k <= NC / 2 ? k : NC - k;
    cpx xk, xnk;
    unmix_fwd({zl[2 * kk], zl[2 * kk + 1]}, {zl[2 * (NC - kk)], zl[2 * (NC - kk) + 1]},
              {st[2 * (kk - 1)], st[2 * (kk - 1) + 1]}, xk, xnk);
    return (k < NC / 2) ? xk : xnk; // k == NC/2: the X[ncfft-k] form is written last in kissfft
}

// x = x(partner) * sgn + x inside a row of 16 lanes as ONE instruction: v_fmac_f32 with the exchange as its DPP operand -- no
// LDS round trip, no separate move (the compiler keeps v_mov_b32_dpp + v_fmac apart when given the builtin).  Partners:
// lane ^ 8 (row_ror:8), lane ^ 7 (row_half_mirror), lane ^ 2 and lane ^ 1 (quad_perm).  There is no single DPP pattern for
// lane ^ 4, so the kernel numbers a frame's rows such that index bit 2 flips with lane bits 0..2 together (see `a` there).
// s_nop 1: a DPP read of a VGPR needs two wait states after the VALU write of it.
__device__ __forceinline__ void dpp_butterfly(float& x, float sgn, int h) // h is a constant after unrolling
{
    switch (h) {
    case 8: asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    case 4: asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break; // lane ^ 7
    case 2: asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    default: asm("s_nop 1\n\tv_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    }
}
// The same without the wait states, for a BLOCK of butterflies on distinct registers between two scheduling fences: one
// `s_nop 1` in front of the block covers the first, and no butterfly reads what its predecessor wrote (dpp_block_begin / _end)
__device__ __forceinline__ void dpp_butterfly_raw(float& x, float sgn, int h)
{
    switch (h) {
    case 8: asm volatile("v_fmac_f32_dpp %0, %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    case 4: asm volatile("v_fmac_f32_dpp %0, %0, %1 row_half_mirror row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    case 2: asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    default: asm volatile("v_fmac_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(sgn)); break;
    }
}
__device__ __forceinline__ void dpp_block_begin()
{
    __builtin_amdgcn_sched_barrier(0); // nothing is scheduled across: the block holds the butterflies only
    asm volatile("s_nop 1");           // a DPP read of a VGPR needs two wait states after the VALU write of it
}
__device__ __forceinline__ void dpp_block_end() { __builtin_amdgcn_sched_barrier(0); }

// dispatch on the wavefront sizes of the N-point transforms
#define VADFFT_DISPATCH(n, CALL)            \
    switch (n) {                            \
    case 512: { constexpr int R = 4; CALL; break; }   \
    case 1024: { constexpr int R = 8; CALL; break; }  \
    case 2048: { constexpr int R = 16; CALL; break; } \
    default: break;                         \
    }


// the generic mixed-radix kernels' launcher (kernels_fftgen.hip), used by K4's and the batch launchers for other sizes
int fvad_launch_rfft_generic_any(const float* frames, long n_frames, const float* window, VadFftPlan pl, float* bins, float* mag,
                                 const VadFftJob* jobs, int n_jobs, long max_frames, int min_bin, int max_bin, hipStream_t stream);
