// kernels_fft.hip -- the spectral front end on gfx950.
//
//   K1 stft_kernel    chunk RMS (BufferedVolumeAnalyzer.zig:48-69, audio_utils.zig:14-24),
//                     /3 decimation (resample.zig:9-29), sqrt-Hann 320-point real FFT
//                     (NSNet2.zig:239-264 -> FFT.zig:85-113), log-power features (NSNet2.zig:266-287)
//   K3 istft_kernel   gain (NSNet2.zig:289-310), inverse real FFT + window + overlap-add
//                     (NSNet2.zig:312-339), x3 linear upsample (resample.zig:32-79)
//   K4 vadfft_kernel  periodic-Hann 1024-point real FFT, |X| * norm, band sum
//                     (BufferedFFT.zig:162-202)
//   rfft/irfft batch  FFT.fft / FFT.invFft for many frames (BASELINE config 2)
//
// All FFTs share one wavefront-level scheme.  A real transform of length 2N is one complex
// transform of length N = R * L over z[n] = x[2n] + i x[2n+1] plus the same un-mixing pass
// kissfft uses ("super twiddles").  The complex transform keeps R points per lane in registers
// and spreads L points over lanes (N = 160: R = 5, L = 32, two frames per 64-lane wavefront;
// N = 512: R = 8, L = 64):
//     X[k1 + R k2] = sum_p W_L^{p k2} ( W_N^{p k1} sum_j z[p + L j] W_R^{j k1} )
// i.e. an R-point DFT in registers, one twiddle multiply, then R independent L-point
// decimation-in-frequency FFTs whose butterflies are lane exchanges (__shfl_xor).  Twiddles are
// read once per wavefront from tables the host evaluated in double (as kissfft does) and kept in
// registers; window coefficients likewise.  f32 throughout; fp contraction is off for this file
// so products and sums round exactly where the reference's do.

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

// ============================================================================ K1
constexpr int K1_THREADS = 256; // the threads that load, decimate and sum (the sample -> thread assignment fixes the RMS bits)
constexpr int K1_BLOCK = 512;   // eight wavefronts share the decimated chunk in LDS for the frame loop
constexpr int K1_DEC = (kRowsPerChunk + 1) * kNHop; // 8800 decimated samples: frames -4..49 (indexed from frame -4; since the
                                                     // warm-up frames are no longer transformed here only [640, 8800) is used)

// parts == 1: one workgroup per chunk does everything.  parts > 1 (launches of a few chunks, where a chunk's 27 frame
// pairs on one workgroup are a latency chain): blockIdx.y < parts transforms its share of the frame pairs from its share
// of the samples; blockIdx.y == parts streams the whole chunk for the RMS (the sample -> thread assignment and the order
// of that sum do not change) and writes the carries.  Every value is computed by the same instructions either way.
__global__ __launch_bounds__(K1_BLOCK) void stft_kernel(const ChunkDesc* __restrict__ descs,
                                                          FftTables tb, float* __restrict__ feat,
                                                          float* __restrict__ spec, int parts)
{
    __shared__ __attribute__((aligned(16))) float dec[K1_DEC];
    __shared__ __attribute__((aligned(16))) float zb[K1_BLOCK / 64][2][2 * 160];
    __shared__ __attribute__((aligned(8))) float s_win[kNFft];
    __shared__ __attribute__((aligned(8))) float s_sth[2 * 81]; // un-mixing table / 2, entry k for bin k (unmix_fwd_h)
    __shared__ float s_red[4];

    const int g = blockIdx.x;
    const ChunkDesc d = descs[g];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int N_PAIRS = kRowsPerChunk / 2; // 27
    const int part = blockIdx.y;
    const bool do_fft = parts == 1 || part < parts;
    const bool do_rms = parts == 1 || part == parts;
    // Frame pairs 0 and 1 are the chunk's four warm-up rows = the last four frames of the PREVIOUS chunk of the lane.  They are
    // not transformed again here: the workgroup of the previous chunk writes its frames 50..53 into this chunk's rows 0..3 as
    // well (below), and a lane's first chunk of a launch takes them from the carry -- 50 transforms per chunk, not 54, and
    // 480 history samples instead of 2400.  The same values either way (they were the same instructions on the same samples).
    constexpr int P_LO = kWarmupRows / 2, P_OWN = N_PAIRS - P_LO; // 2, 25
    const int pa = parts == 1 ? P_LO : (do_fft ? P_LO + (P_OWN * part) / parts : P_LO);          // this workgroup's frame pairs
    const int pb = parts == 1 ? N_PAIRS : (do_fft ? P_LO + (P_OWN * (part + 1)) / parts : P_LO);

    for (int i = tid; i < kNFft; i += K1_BLOCK) s_win[i] = tb.win320[i];
    for (int i = tid; i < 162; i += K1_BLOCK) s_sth[i] = i >= 2 ? tb.st320[i - 2] * 0.5f : (i == 0 ? 0.0f : -0.5f);

    // ---- load + decimate + sum of squares: wavefronts 0..3 (the other four wait at the barrier below; with two workgroups
    // per CU their SIMD slots go to the other workgroup's frame loop meanwhile)
    const bool loader = wave < K1_THREADS / 64;
    float ss = 0.0f;
    if (loader) {
        // [2400 raw samples of history | chunk]: the history of the first chunk of a launch is not in memory (its
        // decimated tail comes from the carry below), so its float4s are skipped -- but the sample -> thread
        // assignment is the SAME for every chunk, first or not: the order of the RMS sum, and with it the RMS bits,
        // must not depend on where a launch or a push happens to start
        constexpr int hist = (kWarmupRows + 1) * kNHop * kDown; // 2400
        const int dec0 = (kWarmupRows + 1) * kNHop;             // 800: where the chunk's own decimated samples start
        // float4s this workgroup needs: all of them for the RMS; for frame pairs [pa, pb) the decimated samples
        // [320 pa, 320 pb + 160), i.e. raw samples [960 pa, 960 pb + 480)
        // (the RMS workgroup starts at the 480 history samples frame 4 reaches back to: float4 480 = 240 P_LO)
        const int i4_lo = 240 * (do_rms ? P_LO : pa), i4_hi = do_rms ? (hist + kChunk48) / 4 : 240 * pb + 120;
        const int i4_begin = (d.first && hist / 4 > i4_lo) ? hist / 4 : i4_lo;
        // float4 i4 of [history | chunk] holds samples 4 i4 .. 4 i4 + 3; the decimated ones are those at multiples of 3:
        // sample 3 q with q = ceil(4 i4 / 3), i.e. element r = 3 q - 4 i4 (0, 1 or 2), and element 3 too when r == 0.
        // (one division per float4 instead of one per sample; the order of the RMS sum is untouched)
        auto take4 = [&](int i4, const f32x4& v) {
            const unsigned x = 4u * (unsigned)i4;
            const unsigned q = (x + 2u) / 3u;
            const unsigned r = 3u * q - x;
            dec[q] = r == 0 ? v.x : (r == 1 ? v.y : v.z);
            if (r == 0) dec[q + 1] = v.w;
            if (i4 >= hist / 4) {
                ss += v.x * v.x;
                ss += v.y * v.y;
                ss += v.z * v.z;
                ss += v.w * v.w;
            }
        };
        // batches of 9 independent loads per thread are issued before any is consumed, so the chunk's
        // 96-105 KB stream in with ~37 KB per workgroup in flight instead of one L2/HBM round trip per loop
        // iteration.  PCM16 input takes the same path with 8-byte loads (4 samples), converted exactly like the
        // host decode; the sample -> thread assignment, and with it the order of the RMS sum, is the same for
        // both formats, so the two give bit-identical results.
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const f32x4* src4 = reinterpret_cast<const f32x4*>(d.in - hist);
        const s16x4* src16 = reinterpret_cast<const s16x4*>(d.in16 - hist);
        const bool pcm16 = d.in16 != nullptr;
        const int n4 = i4_hi;
        constexpr int LD_BATCH = 9;
        for (int base = (i4_lo / (LD_BATCH * K1_THREADS)) * (LD_BATCH * K1_THREADS); base < n4; base += LD_BATCH * K1_THREADS) {
            f32x4 v[LD_BATCH];
            if (pcm16) {
                s16x4 r[LD_BATCH];
#pragma unroll
                for (int b = 0; b < LD_BATCH; ++b) {
                    const int i4 = base + b * K1_THREADS + tid;
                    r[b] = (i4 < n4 && i4 >= i4_begin) ? src16[i4] : (s16x4){0, 0, 0, 0};
                }
#pragma unroll
                for (int b = 0; b < LD_BATCH; ++b)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[b][e] = (float)r[b][e] * (1.0f / 32768.0f);
            } else {
#pragma unroll
                for (int b = 0; b < LD_BATCH; ++b) {
                    const int i4 = base + b * K1_THREADS + tid;
                    v[b] = (i4 < n4 && i4 >= i4_begin) ? src4[i4] : (f32x4){0.f, 0.f, 0.f, 0.f};
                }
            }
#pragma unroll
            for (int b = 0; b < LD_BATCH; ++b) {
                const int i4 = base + b * K1_THREADS + tid;
                if (i4 < n4 && i4 >= i4_begin) take4(i4, v[b]);
            }
        }
        if (d.last && do_rms) {
            for (int j = tid; j < kNHop * kDown; j += K1_THREADS) {
                const int i = kChunk48 - kNHop * kDown + j;
                d.carry_out->in_tail[j] = pcm16 ? (float)d.in16[i] * (1.0f / 32768.0f) : d.in[i];
            }
        }
        if (d.first) {
            // audio_input[0..160) of the reference = decimated tail of the previous call
            const float* tail = d.carry_in->in_tail;
            for (int j = tid; j < kNHop; j += K1_THREADS) dec[dec0 - kNHop + j] = tail[kDown * j];
        }
    }
    ss = wave_sum(ss);
    if (loader && lane == 0) s_red[wave] = ss;
    __syncthreads();
    if (tid == 0 && do_rms) {
        const float sum = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
        if (d.rms) *d.rms = sqrtf(sum / (float)kChunk48);
    }

    // ---- warm-up feature rows of the first chunk of a call come from the carry (zeros at t=0:
    // NSNet2.zig:77-79)
    float* feat_g = feat + (size_t)g * kRowsPerChunk * kFeatStride;
    if (d.first && do_rms && loader) {
        const float* ft = d.carry_in->feat_tail;
        for (int i = tid; i < kWarmupRows * kNBins; i += K1_THREADS) {
            const int r = i / kNBins, k = i - r * kNBins;
            feat_g[r * kFeatStride + k] = ft[i];
        }
    }

    // ---- per-lane constants
    const int half = lane >> 5;
    const int p = lane & 31;
    LaneTw<5, 32> tw;
    lane_tw_load<5, 32, false>(tw, tb.tw160, p);
    const int k2 = bitrev_lane<32>(p);
    const float p_min = 1.0f / 1e12f; // std.math.pow(f32, 10, -12), NSNet2.zig:275
    float* spec_g = spec + (size_t)g * kFramesPerChunk * kNBins * 2;
    // stores of the frame loop as buffer stores: the chunk's rows in the resource, the pair's first row in the scalar offset,
    // a lane's bins at constant 32-bit offsets (its half-wavefront's row included) -- no 64-bit address arithmetic per store
    const auto rs_f = __builtin_amdgcn_make_buffer_rsrc(feat_g, 0, kRowsPerChunk * kFeatStride * 4, 0x00020000);
    const auto rs_sp = __builtin_amdgcn_make_buffer_rsrc(spec_g, 0, kFramesPerChunk * kNBins * 2 * 4, 0x00020000);
    const auto rs_fn = __builtin_amdgcn_make_buffer_rsrc(feat_g + kRowsPerChunk * kFeatStride, 0, kWarmupRows * kFeatStride * 4, 0x00020000);
    unsigned vo_f[3], vo_fn[3], vo_s[3], vo_sn[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        const int k = p + 32 * u;
        vo_f[u] = (unsigned)(half * kFeatStride + k) * 4u;
        vo_fn[u] = (unsigned)(half * kFeatStride + 160 - k) * 4u;
        vo_s[u] = (unsigned)(half * kNBins * 2 + 2 * k) * 4u;
        vo_sn[u] = (unsigned)(half * kNBins * 2 + 2 * (160 - k)) * 4u;
    }

    // Every wavefront runs its own frame pairs pa + wave, pa + wave + 8, ... -- lanes 0..31 the pair's first frame, lanes
    // 32..63 the second -- with no workgroup barrier: the complex transform goes through the wavefront's own LDS slab
    // (LDS accesses of one wavefront execute in program order), and wavefronts drift apart and cover each other's waits.
    for (int pi = pa + wave; pi < pb; pi += K1_BLOCK / 64) {
        const int fl = 2 * pi + half;
        float* z = zb[wave][half];
        {
            cpx v[5];
            const float* x = dec + kNHop * fl;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = 2 * (p + 32 * j);
                const float2 xv = *reinterpret_cast<const float2*>(x + n);
                const float2 wv = *reinterpret_cast<const float2*>(s_win + n);
                v[j] = {xv.x * wv.x, xv.y * wv.y}; // loadSamplesFwd, FFT.zig:183-199
            }
            wave_fft<5, 32, false>(v, tw, p);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k1 = 0; k1 < 5; ++k1) {
                const int k = k1 + 5 * k2;
                *reinterpret_cast<float2*>(z + 2 * k) = make_float2(v[k1].r, v[k1].i);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // un-mix: bins k = p, p + 32, p + 64 (<= 80) and their mirrors 160 - k.  Bin 80 is its own mirror: only the
        // X[ncfft - k] form is kept, the one kissfft writes last.
        const unsigned so_f = (unsigned)(2 * pi) * (kFeatStride * 4);
        const unsigned so_s = (unsigned)(2 * pi - kWarmupRows) * (kNBins * 2 * 4); // used for fl >= kWarmupRows only (whole pairs)
        float* ftail = d.carry_out->feat_tail + (fl - kFramesPerChunk) * kNBins; // used for the lane's last chunk, fl >= 50
        const bool to_spec = fl >= kWarmupRows, to_tail = d.last && fl >= kFramesPerChunk;
        const bool to_next = !d.last && fl >= kFramesPerChunk; // rows 0..3 of the lane's next chunk (chunk g + 1 of the launch)
        const unsigned so_n = (unsigned)(2 * pi - kFramesPerChunk) * (kFeatStride * 4);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const int k = p + 32 * u;
            if (u < 2 || k <= 80) {
                const int kn = 160 - k;
                const bool dc = u == 0 && k == 0;
                const int ksrc = dc ? 0 : kn; // z[160] does not exist: k = 0 pairs with itself (table entry 0 = (0, -1/2))
                const float2 zk = *reinterpret_cast<const float2*>(z + 2 * k);
                const float2 zn = *reinterpret_cast<const float2*>(z + 2 * ksrc);
                const float2 st = *reinterpret_cast<const float2*>(s_sth + 2 * k);
                cpx xk, xnk;
                unmix_fwd_h({zk.x, zk.y}, {zn.x, zn.y}, {st.x, st.y}, xk, xnk);
                if (dc) { xk.i = 0.0f; xnk.i = 0.0f; } // kissfft's DC / Nyquist bins are real: +0, not the formula's -0
                const float fk = log10_pos(fmaxf(xk.r * xk.r + xk.i * xk.i, p_min));
                const float fnk = log10_pos(fmaxf(xnk.r * xnk.r + xnk.i * xnk.i, p_min));
                if (k != 80) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fk), rs_f, vo_f[u], so_f, 0);
                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fnk), rs_f, vo_fn[u], so_f, 0);
                if (to_spec) { // one 8-byte store per bin (rows start 8-byte aligned: 161 * 2 floats per row)
                    if (k != 80) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2v){xk.r, xk.i}), rs_sp, vo_s[u], so_s, 0);
                    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2v){xnk.r, xnk.i}), rs_sp, vo_sn[u], so_s, 0);
                }
                if (to_tail) {
                    if (k != 80) ftail[k] = fk;
                    ftail[kn] = fnk;
                }
                if (to_next) {
                    if (k != 80) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fk), rs_fn, vo_f[u], so_n, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, fnk), rs_fn, vo_fn[u], so_n, 0);
                }
            }
        }
    }
}

// parts: 1, or 2 / 3 for launches of a few chunks (parts + 1 workgroups per chunk)
void fvad_launch_stft(const ChunkDesc* descs, int n_chunks, FftTables tb, float* feat, float* spec,
                      hipStream_t stream, int parts)
{
    if (parts < 1 || parts > 3) parts = 1;
    hipLaunchKernelGGL(stft_kernel, dim3(n_chunks, parts == 1 ? 1 : parts + 1), dim3(K1_BLOCK), 0, stream, descs, tb, feat,
                       spec, parts);
}

// ============================================================================ K3
// One WAVEFRONT per run of output hops, no workgroup barrier in the frame loop.  A chunk's 50 hops are 25 pairs; the
// 4 * parts wavefronts of its workgroups take contiguous runs of pairs [P0, P1).  A wavefront walks its run one frame pair
// (2 pi, 2 pi + 1) at a time -- lanes 0..31 the first frame, lanes 32..63 the second, as in K1 -- and keeps everything
// of a pair in its own 5 KB of LDS: gain * X pre-mixed into the complex sequence, the inverse transform read from and written
// back over it (windowed), then the pair's two output hops formed where they are consumed:
//     d[160 f + j] = y_{f-1}[160 + j] + y_f[j]   (NSNet2.zig:336),
// the second half of the PREVIOUS pair's second frame being in the wavefront's other buffer (two buffers, alternating).
// A run starts one pair early (frames 2 P0 - 2, 2 P0 - 1: nothing written, they leave y_{2 P0 - 1} and the sample before
// the run's first); the chunk's very first pair takes both from the lane's carry instead.  Frames < 0 belong to the previous
// chunk of the same lane (g - 1).  Every value is computed by the same instructions whatever the split: the same bits for
// parts = 1, 2, 3, and the same bits as the round-3 kernel (one workgroup per chunk, 52 frames between two barriers per
// 8 frames, 79 KB of LDS: two workgroups per CU, 46 % of its VALU-issue time busy) at 55 transformed frames instead of 52 (58 without the
// seam sharing below).
// The spectrogram / gain operands of the next pair are fetched as soon as the pre-mix has consumed this pair's.
__global__ __launch_bounds__(256) void istft_kernel(const ChunkDesc* __restrict__ descs, FftTables tb,
                                                    const float* __restrict__ spec,
                                                    const float* __restrict__ gains,
                                                    int g_rows, int g_row0, int n_runs, int n_chunks)
{
    __shared__ __attribute__((aligned(16))) float slab[4][2][2 * kNFft]; // [wavefront][buffer][frame of the pair][320]
    __shared__ __attribute__((aligned(8))) float s_wn[kNFft];
    __shared__ __attribute__((aligned(8))) float s_st[2 * 80];
    // a run's first pair (frames 2 P0 - 2, 2 P0 - 1: its seam) is the LAST pair of the run before it.  Wavefronts 1..3 of a
    // workgroup hold the runs that follow wavefronts 0..2 (4 | n_runs), so they transform their seam pair into a shared buffer
    // and raise a flag, and the wavefront before them takes its last pair from there instead of transforming it again:
    // 55 instead of 58 transformed frames per chunk (parts = 1), three of four seam pairs fetched once instead of twice.
    // A taker only ever waits for a wavefront of its own workgroup that waits for nobody.
    __shared__ __attribute__((aligned(16))) float seam[4][2 * kNFft];
    __shared__ int seam_flag[4];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wavefront w of the launch takes run w % n_runs of chunk w / n_runs (n_runs = 4, 8, 12: fvad_launch_istft)
    const int w_all = 4 * (int)blockIdx.x + wave;
    const int g = w_all / n_runs, run = w_all - g * n_runs;
    const bool idle = g >= n_chunks; // the last workgroup of a launch whose wavefront count is not a multiple of four
    const ChunkDesc d = descs[idle ? 0 : g];
    const int half = lane >> 5;
    const int p = lane & 31;

    for (int i = tid; i < kNFft; i += 256) s_wn[i] = tb.win320n[i];
    for (int i = tid; i < 160; i += 256) s_st[i] = tb.st320[i];
    if (tid < 4) seam_flag[tid] = 0;
    __syncthreads(); // the only workgroup barrier: window and un-mixing table, the seam flags

    LaneTw<5, 32> tw;
    lane_tw_load<5, 32, true>(tw, tb.tw160, p);
    const int k2 = bitrev_lane<32>(p);

    constexpr int HOP_PAIRS = kFramesPerChunk / 2; // 25
    const int P0 = (HOP_PAIRS * run) / n_runs, P1 = (HOP_PAIRS * (run + 1)) / n_runs;
    if (idle || P0 >= P1) return;
    const bool from_carry = d.first && P0 == 0;
    float* buf0 = slab[wave][0];
    float* buf1 = slab[wave][1];
    float dm1 = 0.0f; // decimated output sample just before the pair at hand
    if (from_carry) { // y_{-1}[160..320) and the sample before the chunk come from the lane's carry
        for (int j = lane; j < kNHop; j += 64) buf1[kNFft + kNHop + j] = d.carry_in->ola_tail[j];
        dm1 = d.carry_in->last_sample;
    }

    struct Item { float sk_r, sk_i, snk_r, snk_i, gk, gnk; };
    // bins k = p, p + 32, p + 64 (<= 80) and their mirrors 160 - k of this half-wavefront's frame of pair pi.  Buffer loads: a
    // pair's two frames are consecutive rows of ONE chunk, so the chunk's base goes into the resource (scalar), the pair's row
    // into the scalar offset, and a lane's share is six constant 32-bit offsets -- no 64-bit address arithmetic per load
    unsigned vo_s[3], vo_sn[3], vo_g[3], vo_gn[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {
        int k = p + 32 * u;
        k = k <= 80 ? k : 80; // lanes past bin 80 repeat it: loads stay unconditional, the values are not used
        vo_s[u] = (unsigned)(half * kNBins * 2 + 2 * k) * 4u;
        vo_sn[u] = (unsigned)(half * kNBins * 2 + 2 * (160 - k)) * 4u;
        vo_g[u] = (unsigned)(half * kFeatStride + k) * 4u;
        vo_gn[u] = (unsigned)(half * kFeatStride + 160 - k) * 4u;
    }
    auto fetch = [&](int pi, Item (&itm)[3]) {
        const int gg = pi < 0 ? g - 1 : g;                          // pi = -1: the previous chunk's last two frames
        const int f0 = pi < 0 ? 2 * pi + kFramesPerChunk : 2 * pi;  // the pair's first frame within that chunk
        const auto rs_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(spec) + (size_t)gg * kFramesPerChunk * kNBins * 2, 0,
                                                            kFramesPerChunk * kNBins * 2 * 4, 0x00020000);
        const auto rs_g = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gains) + (size_t)gg * g_rows * kFeatStride, 0,
                                                            g_rows * kFeatStride * 4, 0x00020000);
        const unsigned so_s = (unsigned)f0 * (kNBins * 2 * 4), so_g = (unsigned)(g_row0 + f0) * (kFeatStride * 4);
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const f32x2v a = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(rs_s, vo_s[u], so_s, 0));
            const f32x2v b = __builtin_bit_cast(f32x2v, __builtin_amdgcn_raw_buffer_load_b64(rs_s, vo_sn[u], so_s, 0));
            const float gk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, vo_g[u], so_g, 0));
            const float gnk = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_g, vo_gn[u], so_g, 0));
            itm[u] = Item{a.x, a.y, b.x, b.y, gk, gnk};
        }
    };
    const int pi_begin = from_carry ? 0 : P0 - 1;
    // (only while a workgroup's four wavefronts are four consecutive runs of ONE chunk: 4 | n_runs, which fvad_launch_istft guarantees)
    const bool share = (n_runs & 3) == 0;
    const bool give = share && wave > 0;  // (run > 0 then: never from_carry) this run's seam pair goes to seam[wave] for the wavefront before
    const bool take = share && wave < 3;  // this run's last pair comes from seam[wave + 1] (the next wavefront: same chunk, next run)
    const int P1c = take ? P1 - 1 : P1; // pairs this wavefront transforms: pi_begin .. P1c - 1
    Item cur[3];
    fetch(pi_begin, cur);
    // output: a pair's 960 samples start at a scalar offset of the chunk; a lane's three float4s at constant offsets
    const auto rs_o = __builtin_amdgcn_make_buffer_rsrc(d.den, 0, kChunk48 * 4, 0x00020000);
    const float frac1 = 1.0f / 3.0f, frac2 = 2.0f / 3.0f;
    float* prev = from_carry ? buf1 : nullptr; // the previous pair's buffer (from_carry: the carry's tail)
    for (int pi = pi_begin; pi < P1; ++pi) {
        const bool taken = pi >= P1c, given = give && pi == pi_begin;
        float* cb = given ? seam[wave] : (taken ? seam[wave + 1] : (prev == buf0 ? buf1 : buf0)); // this pair's buffer
        float* pb = prev;
        prev = cb;
        __builtin_amdgcn_wave_barrier();
        if (taken) { // transformed by the next wavefront as its first pair, long ago
            volatile int* fl = seam_flag;
            while (fl[wave + 1] == 0) __builtin_amdgcn_s_sleep(1);
        } else {
        {   // pre-mix gain * X into the length-160 complex sequence
            float* z = cb + half * kNFft;
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int k = p + 32 * u;
                if (u < 2 || k <= 80) {
                    const int kn = 160 - k;
                    float gk = cur[u].gk, gnk = cur[u].gnk;
                    gk = gk < -80.0f ? -80.0f : (gk > 1.0f ? 1.0f : gk);   // NSNet2.zig:295-305
                    gnk = gnk < -80.0f ? -80.0f : (gnk > 1.0f ? 1.0f : gnk);
                    const cpx yk = {cur[u].sk_r * gk, cur[u].sk_i * gk};
                    const cpx ynk = {cur[u].snk_r * gnk, cur[u].snk_i * gnk};
                    if (u == 0 && k == 0) {
                        *reinterpret_cast<float2*>(z) = make_float2(yk.r + ynk.r, yk.r - ynk.r);
                    } else {
                        cpx tk, tnk;
                        const float2 st = *reinterpret_cast<const float2*>(s_st + 2 * (k - 1));
                        premix_inv(yk, ynk, {st.x, -st.y}, tk, tnk);
                        *reinterpret_cast<float2*>(z + 2 * k) = make_float2(tk.r, tk.i);
                        *reinterpret_cast<float2*>(z + 2 * kn) = make_float2(tnk.r, tnk.i); // k == 80: second write wins
                    }
                }
            }
        }
        // the next pair's operands, into the registers the pre-mix has just consumed: in flight during this pair's transform and output
        if (pi + 1 < P1c) fetch(pi + 1, cur);
        __builtin_amdgcn_wave_barrier();
        {
            float* z = cb + half * kNFft;
            cpx v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float2 zv = *reinterpret_cast<const float2*>(z + 2 * (p + 32 * j));
                v[j] = {zv.x, zv.y};
            }
            wave_fft<5, 32, true>(v, tw, p);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int k1 = 0; k1 < 5; ++k1) {
                const int n = 2 * (k1 + 5 * k2);
                // inv_fft_buffer[i] *= window[i] * (1/320), NSNet2.zig:335 -- over the pair's own z, which is in registers by now
                const float2 wv = *reinterpret_cast<const float2*>(s_wn + n);
                *reinterpret_cast<float2*>(z + n) = make_float2(v[k1].r * wv.x, v[k1].i * wv.y);
            }
        }
        if (given) { // LDS operations of a wavefront execute in order: the flag lands behind the pair
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) *(volatile int*)&seam_flag[wave] = 1;
        }
        }
        __builtin_amdgcn_wave_barrier();
        // the pair's 320 decimated samples d[m] = (second half of the earlier frame) + (first half of the later one):
        //   m < 160: y_{2 pi - 1}[160 + m] (other buffer) + y_{2 pi}[m];   m >= 160: y_{2 pi}[m] + y_{2 pi + 1}[m - 160]
        auto dec4 = [&](int m0) -> f32x4 { // 160 % 4 == 0: a float4 never straddles the two hops
            const float* a = m0 < kNHop ? pb + kNFft + kNHop + m0 : cb + m0;
            const float* b = m0 < kNHop ? cb + m0 : cb + kNHop + m0;
            return *reinterpret_cast<const f32x4*>(a) + *reinterpret_cast<const f32x4*>(b);
        };
        auto dec1 = [&](int m) -> float {
            return m < kNHop ? pb[kNFft + kNHop + m] + cb[m] : cb[m] + cb[kNHop + m];
        };
        if (pi >= P0) {
            // x3 upsample (resample.zig:32-79): out[3m+2] = d[m]; out[3m+j] = lerp(d[m-1], d[m], (j+1)/3).
            // Each lane turns 4 decimated samples into 12 outputs = three float4 stores (d.den is 16-byte
            // aligned: chunk regions are 64-float aligned and 24000 % 4 == 0): 80 such items per pair.
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) {
                const int w = lane + 64 * pass;
                if (pass == 0 || w < 2 * kNHop / 4) {
                    const int m0 = 4 * w;
                    const f32x4 c = dec4(m0);
                    const float pv = w > 0 ? dec1(m0 - 1) : dm1;
                    // std.math.lerp = mulAdd: (b - a) * t + a, fused
                    const f32x4 o0 = {__builtin_fmaf(c.x - pv, frac1, pv), __builtin_fmaf(c.x - pv, frac2, pv), c.x,
                                      __builtin_fmaf(c.y - c.x, frac1, c.x)};
                    const f32x4 o1 = {__builtin_fmaf(c.y - c.x, frac2, c.x), c.y, __builtin_fmaf(c.z - c.y, frac1, c.y),
                                      __builtin_fmaf(c.z - c.y, frac2, c.y)};
                    const f32x4 o2 = {c.z, __builtin_fmaf(c.w - c.z, frac1, c.z), __builtin_fmaf(c.w - c.z, frac2, c.z), c.w};
                    const int wo = (2 * kNHop / 4) * pi + w; // float4-triple index within the chunk
                    const unsigned so_o = (unsigned)pi * (2 * kNHop * kDown * 4);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o0), rs_o, (unsigned)w * 48u, so_o, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o1), rs_o, (unsigned)w * 48u + 16u, so_o, 0);
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o2), rs_o, (unsigned)w * 48u + 32u, so_o, 0);
                    if (d.den16) { // PCM16 copy of the same 12 samples: three 8-byte stores
                        typedef short s16x4 __attribute__((ext_vector_type(4)));
                        auto q = [](float y) { return (short)__builtin_rintf(fminf(fmaxf(y * 32768.0f, -32768.0f), 32767.0f)); };
                        s16x4* o16 = reinterpret_cast<s16x4*>(d.den16) + 3 * wo;
                        o16[0] = (s16x4){q(o0.x), q(o0.y), q(o0.z), q(o0.w)};
                        o16[1] = (s16x4){q(o1.x), q(o1.y), q(o1.z), q(o1.w)};
                        o16[2] = (s16x4){q(o2.x), q(o2.y), q(o2.z), q(o2.w)};
                    }
                }
            }
        }
        dm1 = dec1(2 * kNHop - 1);
        if (d.last && pi == HOP_PAIRS - 1) { // the lane's carry: y_49[160..320) and the chunk's last decimated sample
            for (int j = lane; j < kNHop; j += 64) d.carry_out->ola_tail[j] = cb[kNFft + kNHop + j];
            if (lane == 0) d.carry_out->last_sample = dm1;
        }
    }
}

// parts: 1, or 2 / 3 for launches of a few chunks (as K1): 4 * parts runs of hop pairs per chunk.  (Two runs or one per chunk
// at large launches -- fewer seams: 54 or 52 transformed frames instead of 58, 8 % fewer bytes fetched -- were measured on one
// box, alternating: 0.68 and 0.71 ms against 0.67 ms at 16384 chunks; the longer runs overlap worse.)
void fvad_launch_istft(const ChunkDesc* descs, int n_chunks, FftTables tb, const float* spec,
                       const float* gains, int gains_rows_per_chunk, int gains_row0,
                       hipStream_t stream, int parts)
{
    if (parts < 1 || parts > 3) parts = 1;
    const int n_runs = 4 * parts;
    const long waves = (long)n_chunks * n_runs;
    hipLaunchKernelGGL(istft_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, stream, descs, tb, spec, gains,
                       gains_rows_per_chunk, gains_row0, n_runs, n_chunks);
}

// ============================================================================ K4 / rfft-N, N = 128 R
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

template <int R>
__device__ __forceinline__ void vadfft_body(const float* __restrict__ den, long n_frames, long frame,
                                            const VadFftPlan& pl, int min_bin, int max_bin,
                                            float* __restrict__ band_sum, float* __restrict__ bins_out,
                                            float (*zl)[128 * R], float (*mag)[64])
{
    constexpr int N = 128 * R, NB = N / 2 + 1;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const bool active = frame < n_frames;
    if (active) rfftN_wave<R>(den + frame * N, pl.win, pl, zl[wave], lane);
    __syncthreads();
    const float norm = pl.norm;
    if (active && bins_out) {
        for (int k = lane; k < NB; k += 64) {
            const cpx xk = rfftN_bin<R>(zl[wave], pl.st, k);
            bins_out[frame * NB + k] = sqrtf(xk.r * xk.r + xk.i * xk.i) * norm; // FFT.zig:16-18
        }
    }
    const int nb = max_bin - min_bin + 1;
    // band bins (<= 64 of them per pass), then the reference's index-order sum in one lane
    float acc = 0.0f;
    for (int base = 0; base < nb; base += 64) {
        const int k = min_bin + base + lane;
        if (active && base + lane < nb) {
            const cpx xk = rfftN_bin<R>(zl[wave], pl.st, k);
            mag[wave][lane] = sqrtf(xk.r * xk.r + xk.i * xk.i) * norm;
        }
        __syncthreads();
        if (active && lane == 0) {
            const int cnt = nb - base < 64 ? nb - base : 64;
            for (int i = 0; i < cnt; ++i) acc += mag[wave][i]; // BufferedFFT.zig:192-199
        }
        __syncthreads();
    }
    if (active && lane == 0) band_sum[frame] = acc;
}

template <int R>
__global__ __launch_bounds__(256) void vadfft_kernel(const float* __restrict__ den, long n_frames,
                                                     VadFftPlan pl, int min_bin, int max_bin,
                                                     float* __restrict__ band_sum,
                                                     float* __restrict__ bins_out)
{
    __shared__ __attribute__((aligned(16))) float zl[4][128 * R];
    __shared__ float mag[4][64];
    const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    vadfft_body<R>(den, n_frames, frame, pl, min_bin, max_bin, band_sum, bins_out, zl, mag);
}

template <int R>
__global__ __launch_bounds__(256) void vadfft_jobs_kernel(const VadFftJob* __restrict__ jobs, VadFftPlan pl,
                                                          int min_bin, int max_bin)
{
    __shared__ __attribute__((aligned(16))) float zl[4][128 * R];
    __shared__ float mag[4][64];
    const VadFftJob j = jobs[blockIdx.y];
    if ((long)blockIdx.x * 4 >= j.n_frames) return; // whole workgroup past this lane's frames
    const long frame = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    vadfft_body<R>(j.den, j.n_frames, frame, pl, min_bin, max_bin, j.band_sum, j.bins, zl, mag);
}

// ============================================================================ K4 at 1024 points, band sum only: four frames per wavefront
// The kernel above gives a frame to a whole wavefront (8 points per lane, six exchange stages across 64 lanes, all 513 bins)
// and a wavefront to one frame: its twiddles are fetched again for every frame.  The VAD consumes the band min_bin..max_bin
// only (bins 11..43 at the reference's 500-2000 Hz, VADMachine.zig:146-151).  Here a frame is SIXTEEN lanes with 32 complex
// points each, z[32 a + n2] in lane a, four frames per wavefront, wavefronts persistent over their job's frames:
//   Z[k1 + 16 k2] = sum_n2 W512^{n2 k1} W32^{n2 k2} ( sum_a z[32 a + n2] W16^{a k1} )
//   * the inner 16-point transforms run across the lanes (four exchange stages instead of six, none crossing a row of 16:
//     no v_permlane swaps), leaving k1 = bitrev(a) in lane a; one twiddle multiply (31 per lane, loaded once);
//   * the outer 32-point transform is in registers and PRUNED: bins 1..47 and their un-mixing partners 465..511 have
//     k2 in {0, 1, 2} and {29, 30, 31} -- six outputs of 32 (eight 4-point transforms, then six 8-term sums);
//   * kissfft's un-mixing pairs Z[k] with Z[512 - k], which sits in the lane holding 16 - k1: one ds_bpermute per scalar;
//     every lane un-mixes its three bins k1, k1 + 16, k1 + 32, the 48 magnitudes go to LDS and one lane per frame adds the
//     band in index order (BufferedFFT.zig:192-199).
// A lane's 32 points are 256 contiguous bytes, so the four frames are staged through LDS: rows of 64 floats padded to 68 (a
// 16-lane ds_read_b128 then covers all 64 banks).  The staging is LDS-DMA (16 bytes per lane, global address per lane, LDS
// address lane * 16: 17 instructions per four frames, the padding units fetch a neighbour) issued as soon as the previous
// group's points are in registers, so a group's memory time lies under the previous group's arithmetic; a job whose frames
// are not 16-byte aligned takes plain 8-byte loads instead -- the same arithmetic, the same bits.
// ~330 VALU instructions per frame instead of ~650.  The full-spectrum kernel above stays for the magnitude tap, for
// 512 / 2048 points and for bands outside 1..47.
constexpr int V4_ROW = 68; // floats per padded row of 64

// forward twiddle exp(-2 pi i m / 32); constants rounded from double
__device__ __forceinline__ cpx w32(int m)
{
    const float C[9] = {1.0f, 0.98078528040323043f, 0.92387953251128674f, 0.83146961230254524f, 0.70710678118654752f,
                        0.55557023301960218f, 0.38268343236508977f, 0.19509032201612825f, 0.0f};
    m &= 31;
    const int quad = m >> 3, r = m & 7;
    float c, sn;
    switch (quad) {
    case 0: c = C[r]; sn = C[8 - r]; break;
    case 1: c = -C[8 - r]; sn = C[r]; break;
    case 2: c = -C[r]; sn = -C[8 - r]; break;
    default: c = C[8 - r]; sn = -C[r]; break;
    }
    return {c, -sn};
}
__device__ __forceinline__ cpx mul_w32(cpx a, int m) // a * W32^m; the quarter turns are exact
{
    m &= 31;
    if (m == 0) return a;
    if (m == 8) return mul_mi<false>(a);
    if (m == 16) return {-a.r, -a.i};
    if (m == 24) return mul_mi<true>(a);
    return cmul_fma(a, w32(m));
}
// u[n2], n2 < 32, in; y = Y[0], Y[1], Y[2], Y[29], Y[30], Y[31] of the 32-point transform out (u is overwritten).
// n2 = 8 m + s: Y[k2] = sum_s W32^{s k2} T_s[k2 mod 4], T_s[c] = sum_m u[8 m + s] W4^{m c}
__device__ __forceinline__ void dft32_band(cpx (&u)[32], cpx (&y)[6])
{
#pragma unroll
    for (int s = 0; s < 8; ++s) dft4<false>(u[s], u[8 + s], u[16 + s], u[24 + s]); // T_s[c] is now u[8 c + s]
    constexpr int K2[6] = {0, 1, 2, 29, 30, 31};
#pragma unroll
    for (int o = 0; o < 6; ++o) {
        const int k2 = K2[o], c = k2 & 3;
        cpx acc = u[8 * c];
#pragma unroll
        for (int s = 1; s < 8; ++s) acc = cadd(acc, mul_w32(u[8 * c + s], s * k2));
        y[o] = acc;
    }
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

// MINB / MAXB: the band as compile-time constants (the reference's 11..43: the index-order sum is then 33 adds), or 0, 0: the
// band is the run-time pair (a select per bin)
template <int MINB, int MAXB>
__global__ __launch_bounds__(256) void vadfft1024_band_kernel(const VadFftJob* __restrict__ jobs, VadFftPlan pl, int min_bin, int max_bin, int plain_loads)
{
    // dynamic LDS (77 KB: over the static limit; two workgroups per CU): per wavefront a slab of 4 frames x 16 rows of 64 (+4)
    // floats; the window in the same padded rows; 48 magnitudes per frame
    extern __shared__ __attribute__((aligned(16))) float v4_smem[];
    float* s_win = v4_smem + 4 * 64 * V4_ROW;
    float* s_mag = s_win + 16 * V4_ROW;
    const VadFftJob job = jobs[blockIdx.y];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int q = lane >> 4; // frame of the group
    // row of the frame (points 32 a .. 32 a + 31) held by this lane: bits (l3, l2, l1 ^ l2, l0 ^ l2) of the lane index, so that
    // flipping bit 2 of `a` is lane ^ 7 -- a DPP pattern -- and flipping bits 0, 1, 3 stays lane ^ 1, ^ 2, ^ 8
    const int a = (lane & 12) | (((lane & 3) ^ ((lane & 4) ? 3 : 0)));
    const long n_groups = (job.n_frames + 3) / 4;
    long grp = (long)blockIdx.x * 4 + wave;
    const long stride = (long)gridDim.x * 4;
    const bool has_work = grp < n_groups;

    // ---- staging
    float* sl = v4_smem + wave * (64 * V4_ROW);
    const bool dma = (reinterpret_cast<uintptr_t>(job.den) & 15) == 0 && !plain_loads;
    unsigned voff[17]; // 16-byte unit U = 64 jj + lane of the padded slab: row U / 17, unit U % 17 (16 = padding: re-fetches unit 15)
#pragma unroll
    for (int jj = 0; jj < 17; ++jj) {
        const unsigned U = 64u * jj + (unsigned)lane, row = U / 17u, c = U - 17u * row;
        voff[jj] = row * 256u + (c < 16u ? c : 15u) * 16u;
    }
    auto stage = [&](long g0) { // frames 4 g0 .. 4 g0 + 3 -> slab (frames past the job's end read as zeros)
        const long left = job.n_frames - 4 * g0; // > 0
        const unsigned bytes = (unsigned)(left < 4 ? left : 4) * 4096u;
        const float* src = job.den + g0 * 4096;
        if (dma) {
            const auto rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, (int)bytes, 0x00020000);
            auto lds3 = (__attribute__((address_space(3))) char*)sl;
#pragma unroll
            for (int jj = 0; jj < 17; ++jj)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(lds3 + jj * 1024), 16, voff[jj], 0, 0, 0);
        } else {
#pragma unroll 4
            for (int t = 0; t < 32; ++t) {
                const unsigned f = 2u * ((unsigned)lane + 64u * t);
                const float2 v = f * 4u < bytes ? *reinterpret_cast<const float2*>(src + f) : make_float2(0.0f, 0.0f);
                *reinterpret_cast<float2*>(sl + (f >> 6) * V4_ROW + (f & 63u)) = v;
            }
        }
    };
    // the first group's staging goes out before anything else: its memory time lies under the window copy and the twiddle loads
    if (has_work) stage(grp);
    for (int i = tid; i < 1024; i += 256) s_win[(i >> 6) * V4_ROW + (i & 63)] = pl.win[i];
    __syncthreads(); // the only workgroup barrier
    if (!has_work) return;

    // ---- per-lane constants
    const int k1 = (int)(__brev((unsigned)a) >> 28);
    cpx twl[31]; // W512^{n2 k1}
#pragma unroll
    for (int n2 = 1; n2 < 32; ++n2) {
        const cpx t = ld_tw(pl.tw, n2 * k1);
        twl[n2 - 1] = (a & 1) ? cpx{-t.r, -t.i} : t; // the last exchange stage leaves minus the value on odd lanes
    }
    cpx tws[3]; // exchange-stage twiddles (strides 8, 4, 2): 1 on the lower lane, MINUS the twiddle on the upper (mine - other)
#pragma unroll
    for (int st = 0; st < 3; ++st) {
        const int h = 8 >> st;
        cpx t = {1.0f, 0.0f};
        if (a & h) {
            t = ld_tw(pl.tw, (a & (h - 1)) * (256 / h)); // W_{2h}^{a mod h} = W512^{(a mod h) 256 / h}
            t = {-t.r, -t.i};
        }
        tws[st] = t;
    }
    cpx stl[3]; // un-mixing factors of this lane's bins k1 + 16 j (bin 0 is never in the band)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const int k = k1 + 16 * j;
        stl[j] = k > 0 ? ld_tw(pl.st, k - 1) : cpx{0.0f, 0.0f};
    }
    const int a_p = (int)(__brev((unsigned)((16 - k1) & 15)) >> 28); // row whose lane holds 16 - k1
    const int partner = 4 * (16 * q + ((a_p & 12) | ((a_p & 3) ^ ((a_p & 4) ? 3 : 0)))); // its lane, as a byte address for ds_bpermute
    const float norm = pl.norm;

    for (; grp < n_groups; grp += stride) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        cpx u[32];
        {
            const f32x4* xr = reinterpret_cast<const f32x4*>(sl + (16 * q + a) * V4_ROW);
            const f32x4* wr = reinterpret_cast<const f32x4*>(s_win + a * V4_ROW);
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const f32x4 x = xr[t], w = wr[t];
                u[2 * t] = {x.x * w.x, x.y * w.y};
                u[2 * t + 1] = {x.z * w.z, x.w * w.w};
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        if (grp + stride < n_groups) stage(grp + stride); // the slab is in registers: refill it under this group's arithmetic

        // 16-point transforms across the frame's lanes (decimation in frequency, as wave_fft).  Every stage is
        // t = other * sgn + mine with sgn = -1 on the upper lane (there: mine - other, and the stage twiddle is stored
        // negated), the exchange being the DPP operand of that fma (dpp_butterfly).  The last stage has no twiddle: its upper lane holds MINUS the result,
        // which the twiddle multiply that follows absorbs (twl is negated there; u[0] has none and is negated by hand).
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const int h = 8 >> st;
            const float sgn = (a & h) ? -1.0f : 1.0f;
            dpp_block_begin();
#pragma unroll
            for (int n2 = 0; n2 < 32; ++n2) {
                dpp_butterfly_raw(u[n2].r, sgn, h);
                dpp_butterfly_raw(u[n2].i, sgn, h);
            }
            dpp_block_end();
            if (h > 1) {
#pragma unroll
                for (int n2 = 0; n2 < 32; ++n2) u[n2] = cmul_fma(u[n2], tws[st]);
            }
        }
        if (a & 1) u[0] = {-u[0].r, -u[0].i};
#pragma unroll
        for (int n2 = 1; n2 < 32; ++n2) u[n2] = cmul_fma(u[n2], twl[n2 - 1]);
        cpx y[6];
        dft32_band(u, y); // Z[k1 + 16 k2], k2 = 0, 1, 2, 29, 30, 31

        // Z[512 - k] for k = k1 + 16 j: lane of 16 - k1, k2 = 31 - j; for k1 = 0 this lane itself, k2 = 32 - j
        cpx pz[3]; // partner's k2 = 29, 30, 31
#pragma unroll
        for (int o = 0; o < 3; ++o) {
            pz[o].r = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(y[3 + o].r)));
            pz[o].i = __int_as_float(__builtin_amdgcn_ds_bpermute(partner, __float_as_int(y[3 + o].i)));
        }
        const bool self = k1 == 0;
        const cpx zn[3] = {pz[2], self ? pz[2] : pz[1], self ? pz[1] : pz[0]};
        float* mg = s_mag + (4 * wave + q) * 48;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            cpx xk, xnk;
            unmix_fwd(y[j], zn[j], stl[j], xk, xnk);
            mg[k1 + 16 * j] = sqrtf(xk.r * xk.r + xk.i * xk.i) * norm; // FFT.zig:16-18 (bin 0's entry is never read)
        }
        __builtin_amdgcn_wave_barrier();
        if (a == 0) {
            const f32x4* m4 = reinterpret_cast<const f32x4*>(mg);
            float m[48];
#pragma unroll
            for (int t = 0; t < 12; ++t) {
                const f32x4 v = m4[t];
                m[4 * t] = v.x; m[4 * t + 1] = v.y; m[4 * t + 2] = v.z; m[4 * t + 3] = v.w;
            }
            float acc = 0.0f; // index order, BufferedFFT.zig:192-199
            if constexpr (MAXB > 0) {
#pragma unroll
                for (int k = MINB; k <= MAXB; ++k) acc += m[k];
            } else {
#pragma unroll
                for (int k = 1; k < 48; ++k) acc = (k >= min_bin && k <= max_bin) ? acc + m[k] : acc;
            }
            const long frame = 4 * grp + q;
            if (frame < job.n_frames) job.band_sum[frame] = acc;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ============================================================================ any even size: generic mixed radix
// FFT.init takes any even n_fft that kissfft factors (FFT.zig:35-60), VADPipeline.Config.fft_size with it.  The sizes the
// pipeline runs at have wavefront kernels above; every other even size up to kVadFftMax runs here: one workgroup per frame,
// the packed complex transform of length M = n / 2 as Stockham autosort passes through LDS over the radices of M (any
// radix: an output is the direct sum of its R inputs times table twiddles, R complex fmas -- simple rather than fast: a
// boundary-completeness path, not a hot one), then kissfft's un-mixing pass.  Tables (M-th roots, un-mixing factors) are
// evaluated in double on the host like kissfft's.
__device__ __forceinline__ void generic_cfft(cpx* a, cpx* b, int M, const VadFftPlan& pl, bool inverse, cpx*& result)
{
    const int tid = threadIdx.x;
    int Ns = 1;
    cpx* src = a;
    cpx* dst = b;
    for (int f = 0; f < pl.n_fac; ++f) {
        const int R = pl.fac[f];
        const int span = M / R;             // inputs of one output: src[j + r span]
        const int tstep = M / (Ns * R);     // table stride of the pass
        for (int o = tid; o < M; o += 256) {
            const int q = o / span, j = o - q * span;
            const int k = j % Ns;
            const int stride = (int)(((long long)(k + q * Ns) * tstep) % M);
            cpx acc = {0.0f, 0.0f};
            int idx = 0;
            if (R <= 5) {
                for (int r = 0; r < R; ++r) {
                    const cpx x = src[j + r * span];
                    const float wr = pl.tw[2 * idx], wi = inverse ? -pl.tw[2 * idx + 1] : pl.tw[2 * idx + 1];
                    acc.r += x.r * wr - x.i * wi;
                    acc.i += x.r * wi + x.i * wr;
                    idx += stride;
                    if (idx >= M) idx -= M;
                }
            } else {
                // a long direct sum (a prime radix such as 127): accumulated in double, so that its round-off stays at the
                // level of the short butterflies' (kissfft's generic butterfly sums in f32 in another order; both are
                // compared with the oracle at 1e-4 of bins that may be 1e-3 of the frame's largest)
                double ar = 0.0, ai = 0.0;
                for (int r = 0; r < R; ++r) {
                    const cpx x = src[j + r * span];
                    const double wr = pl.tw[2 * idx], wi = inverse ? -pl.tw[2 * idx + 1] : pl.tw[2 * idx + 1];
                    ar += (double)x.r * wr - (double)x.i * wi;
                    ai += (double)x.r * wi + (double)x.i * wr;
                    idx += stride;
                    if (idx >= M) idx -= M;
                }
                acc = {(float)ar, (float)ai};
            }
            dst[(j / Ns) * Ns * R + k + q * Ns] = acc;
        }
        __syncthreads();
        Ns *= R;
        cpx* t = src; src = dst; dst = t;
    }
    result = src;
}

// forward: frame (n samples) x window -> X[0 .. n/2] in `X` (LDS, M + 1 entries)
__device__ __forceinline__ void generic_rfft(const float* __restrict__ x, const float* __restrict__ win, const VadFftPlan& pl, cpx* bufA, cpx* bufB, cpx*& X)
{
    const int M = pl.n / 2;
    for (int j = threadIdx.x; j < M; j += 256) bufA[j] = {x[2 * j] * win[2 * j], x[2 * j + 1] * win[2 * j + 1]};
    __syncthreads();
    cpx* F;
    generic_cfft(bufA, bufB, M, pl, false, F);
    cpx* out = F == bufA ? bufB : bufA;
    // kiss_fftr's un-mixing (k and M - k together; at k == M - k the X[M - k] form is the one written last)
    for (int k = threadIdx.x; k <= M / 2; k += 256) {
        if (k == 0) {
            out[0] = {F[0].r + F[0].i, 0.0f};
            out[M] = {F[0].r - F[0].i, 0.0f};
        } else {
            cpx xk, xnk;
            unmix_fwd(F[k], F[M - k], {pl.st[2 * (k - 1)], pl.st[2 * (k - 1) + 1]}, xk, xnk);
            if (k != M - k) out[k] = xk;
            out[M - k] = xnk;
        }
    }
    __syncthreads();
    X = out;
}

// FFT.fft for many frames (bins and / or magnitudes), and K4's band sum (jobs != nullptr: one lane's frames per blockIdx.y)
__global__ __launch_bounds__(256) void rfft_generic_kernel(const float* __restrict__ frames, long n_frames, const float* __restrict__ window,
                                                           VadFftPlan pl, float* __restrict__ bins, float* __restrict__ mag,
                                                           const VadFftJob* __restrict__ jobs, int min_bin, int max_bin)
{
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int M = pl.n / 2, NB = M + 1;
    cpx* bufA = reinterpret_cast<cpx*>(gsm);
    cpx* bufB = bufA + NB;
    const long frame = blockIdx.x;
    const float* x;
    float* band_sum = nullptr;
    float* bins_mag = nullptr;   // K4's optional |X| norm tap
    if (jobs) {
        const VadFftJob j = jobs[blockIdx.y];
        if (frame >= j.n_frames) return;
        x = j.den + frame * pl.n;
        band_sum = j.band_sum + frame;
        bins_mag = j.bins ? j.bins + frame * NB : nullptr;
        window = pl.win;
    } else {
        if (frame >= n_frames) return;
        x = frames + frame * pl.n;
    }
    cpx* X;
    generic_rfft(x, window, pl, bufA, bufB, X);
    if (!jobs) {
        for (int k = threadIdx.x; k < NB; k += 256) {
            if (bins) { bins[(frame * NB + k) * 2] = X[k].r; bins[(frame * NB + k) * 2 + 1] = X[k].i; }
            if (mag) mag[frame * NB + k] = sqrtf(X[k].r * X[k].r + X[k].i * X[k].i);
        }
        return;
    }
    // |X| norm (FFT.zig:16-18) for the tap and the band, then the reference's index-order sum in one lane (BufferedFFT.zig:192-199)
    float* m = reinterpret_cast<float*>(X == bufA ? bufB : bufA);
    for (int k = threadIdx.x; k < NB; k += 256) {
        const float v = sqrtf(X[k].r * X[k].r + X[k].i * X[k].i) * pl.norm;
        m[k] = v;
        if (bins_mag) bins_mag[k] = v;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float acc = 0.0f;
        for (int k = min_bin; k <= max_bin; ++k) acc += m[k];
        *band_sum = acc;
    }
}

// FFT.invFft: kiss_fftri's pre-mixing, the inverse complex transform, unscaled
__global__ __launch_bounds__(256) void irfft_generic_kernel(const float* __restrict__ bins, long n_frames, VadFftPlan pl, float* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) float gsm[];
    const int M = pl.n / 2, NB = M + 1;
    cpx* bufA = reinterpret_cast<cpx*>(gsm);
    cpx* bufB = bufA + NB;
    const long frame = blockIdx.x;
    if (frame >= n_frames) return;
    const float* b = bins + frame * NB * 2;
    for (int k = threadIdx.x; k <= M / 2; k += 256) {
        const cpx yk = {b[2 * k], b[2 * k + 1]}, ynk = {b[2 * (M - k)], b[2 * (M - k) + 1]};
        if (k == 0) bufA[0] = {yk.r + ynk.r, yk.r - ynk.r};
        else {
            cpx tk, tnk;
            premix_inv(yk, ynk, {pl.st[2 * (k - 1)], -pl.st[2 * (k - 1) + 1]}, tk, tnk);
            if (k != M - k) bufA[k] = tk;
            bufA[M - k] = tnk;
        }
    }
    __syncthreads();
    cpx* T;
    generic_cfft(bufA, bufB, M, pl, true, T);
    float* o = out + frame * pl.n;
    for (int j = threadIdx.x; j < M; j += 256) { o[2 * j] = T[j].r; o[2 * j + 1] = T[j].i; }
}

void fvad_launch_irfft_generic(const float* bins, long n_frames, VadFftPlan pl, float* out, hipStream_t stream)
{
    if (n_frames <= 0) return;
    const size_t lds = (size_t)(pl.n / 2 + 1) * 2 * sizeof(cpx);
    (void)hipFuncSetAttribute((const void*)irfft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(irfft_generic_kernel, dim3((unsigned)n_frames), dim3(256), lds, stream, bins, n_frames, pl, out);
}

static void launch_rfft_generic(const float* frames, long n_frames, const float* window, VadFftPlan pl, float* bins, float* mag,
                                const VadFftJob* jobs, int n_jobs, long max_frames, int min_bin, int max_bin, hipStream_t stream)
{
    const size_t lds = (size_t)(pl.n / 2 + 1) * 2 * sizeof(cpx);
    (void)hipFuncSetAttribute((const void*)rfft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    const dim3 grid((unsigned)(jobs ? max_frames : n_frames), (unsigned)(jobs ? n_jobs : 1));
    hipLaunchKernelGGL(rfft_generic_kernel, grid, dim3(256), lds, stream, frames, n_frames, window, pl, bins, mag, jobs, min_bin, max_bin);
}

#define VADFFT_DISPATCH(n, CALL)            \
    switch (n) {                            \
    case 512: { constexpr int R = 4; CALL; break; }   \
    case 1024: { constexpr int R = 8; CALL; break; }  \
    case 2048: { constexpr int R = 16; CALL; break; } \
    default: break;                         \
    }

void fvad_launch_vadfft(const float* den, long n_frames, VadFftPlan pl, int min_bin, int max_bin,
                        float* band_sum, float* bins_or_null, hipStream_t stream)
{
    if (n_frames <= 0) return;
    if (pl.generic) { // (single-lane form: a one-entry job table would need device memory; the engine uses the jobs form)
        return;
    }
    VADFFT_DISPATCH(pl.n, hipLaunchKernelGGL(vadfft_kernel<R>, dim3((unsigned)((n_frames + 3) / 4)), dim3(256), 0, stream, den,
                                              n_frames, pl, min_bin, max_bin, band_sum, bins_or_null))
}

void fvad_launch_vadfft_jobs(const VadFftJob* jobs, int n_jobs, long max_frames, VadFftPlan pl,
                             int min_bin, int max_bin, hipStream_t stream, int any_bins, int n_cu, int plain)
{
    if (n_jobs <= 0 || max_frames <= 0) return;
    if (pl.generic) {
        launch_rfft_generic(nullptr, 0, nullptr, pl, nullptr, nullptr, jobs, n_jobs, max_frames, min_bin, max_bin, stream);
        return;
    }
    // 1024 points and a band inside bins 1..47 (the reference's 500-2000 Hz is 11..43): the four-frames-per-wavefront
    // kernel writes the band sums; the full-spectrum kernel runs (first) only when some job wants the magnitude tap, so
    // that a call's band sums have the same bits with and without the tap
    const bool band = pl.n == 1024 && min_bin >= 1 && max_bin <= 47 && min_bin <= max_bin;
    if (!band || any_bins)
        VADFFT_DISPATCH(pl.n, hipLaunchKernelGGL(vadfft_jobs_kernel<R>, dim3((unsigned)((max_frames + 3) / 4), (unsigned)n_jobs), dim3(256), 0,
                                                  stream, jobs, pl, min_bin, max_bin))
    if (band) {
        // persistent wavefronts: two workgroups per CU (77 KB of LDS each) over all jobs -- one resident round, no tail --,
        // each wavefront walking its job's groups of four frames with the next group's staging in flight
        const long groups = (max_frames + 3) / 4, wg_all = (groups + 3) / 4;
        long per_job = (2L * (n_cu > 0 ? n_cu : 256)) / n_jobs;
        if (per_job < 1) per_job = 1;
        if (per_job > wg_all) per_job = wg_all;
        constexpr size_t lds = (size_t)(4 * 64 * V4_ROW + 16 * V4_ROW + 16 * 48) * sizeof(float);
        if (min_bin == 11 && max_bin == 43) { // VADMachine.zig:146-151 at 48 kHz / 1024 points
            if (hipFuncSetAttribute((const void*)vadfft1024_band_kernel<11, 43>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return;
            hipLaunchKernelGGL((vadfft1024_band_kernel<11, 43>), dim3((unsigned)per_job, (unsigned)n_jobs), dim3(256), lds, stream, jobs, pl, min_bin, max_bin, plain);
        } else {
            if (hipFuncSetAttribute((const void*)vadfft1024_band_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return;
            hipLaunchKernelGGL((vadfft1024_band_kernel<0, 0>), dim3((unsigned)per_job, (unsigned)n_jobs), dim3(256), lds, stream, jobs, pl, min_bin, max_bin, plain);
        }
    }
}

// ============================================================================ batched FFT.fft
template <int R>
__global__ __launch_bounds__(256) void rfftN_batch_kernel(const float* __restrict__ frames,
                                                          long n_frames,
                                                          const float* __restrict__ window,
                                                          VadFftPlan pl, float* __restrict__ bins,
                                                          float* __restrict__ mag)
{
    constexpr int N = 128 * R, NB = N / 2 + 1;
    __shared__ __attribute__((aligned(16))) float zl[4][N];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const long frame = (long)blockIdx.x * 4 + wave;
    const bool active = frame < n_frames;
    if (active) rfftN_wave<R>(frames + frame * N, window, pl, zl[wave], lane);
    __syncthreads();
    if (!active) return;
    for (int k = lane; k < NB; k += 64) {
        const cpx xk = rfftN_bin<R>(zl[wave], pl.st, k);
        if (bins) { bins[(frame * NB + k) * 2] = xk.r; bins[(frame * NB + k) * 2 + 1] = xk.i; }
        if (mag) mag[frame * NB + k] = sqrtf(xk.r * xk.r + xk.i * xk.i);
    }
}

// 320-point batch (BASELINE config 2).  Every wavefront runs its own pipeline over groups of 4 consecutive frames,
// two at a time (two frames per 64-lane wavefront, as in K1) -- no workgroup barrier anywhere, so wavefronts
// drift apart and cover each other's memory waits.  Per group and wavefront:
//   * input: the 4 frames are 5 KB of contiguous samples; each lane fetches five float4 of the NEXT group into
//     registers before this group's arithmetic and parks them in the wavefront's LDS slab after it (a request
//     has a whole iteration to land);
//   * the complex transform of a frame pair is written over the pair's (consumed) samples in the slab, un-mixed
//     from there, and the 4 x 161 magnitudes (4 x 161 complex bins) are assembled in an LDS tile that leaves as
//     flat float4 stores: a group's output starts at a multiple of 4 x 161 floats, i.e. 16-byte aligned.
// LDS accesses of one wavefront execute in program order, so the slab needs no synchronisation beyond the
// compiler keeping that order (wave_barrier).
constexpr int RB_WF = 4;                        // frames per wavefront iteration
#ifndef RB_OCC
#define RB_OCC 4
#endif

template <bool HAS_BINS>
__global__ __launch_bounds__(256, RB_OCC) void rfft320_batch_kernel(const float* __restrict__ frames,
                                                                    long n_frames,
                                                                    const float* __restrict__ window,
                                                                    FftTables tb, float* __restrict__ bins,
                                                                    float* __restrict__ mag, int vec_ok)
{
    __shared__ __attribute__((aligned(16))) float s_in[4][RB_WF * kNFft];
    __shared__ __attribute__((aligned(16))) float s_mag[4][RB_WF * kNBins];
    __shared__ __attribute__((aligned(16))) float s_bin[4][HAS_BINS ? RB_WF * kNBins * 2 : 4];
    __shared__ __attribute__((aligned(8))) float s_sth[2 * 81]; // un-mixing table / 2, entry k for bin k (unmix_fwd_h)
    __shared__ __attribute__((aligned(8))) float s_win[kNFft];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int half = lane >> 5;
    const int p = lane & 31;
    LaneTw<5, 32> tw;
    lane_tw_load<5, 32, false>(tw, tb.tw160, p);
    const int k2 = bitrev_lane<32>(p);
    // window coefficients are re-read from LDS for every frame pair (5 x ds_read_b64): ten registers less, which
    // is what lets five wavefronts per SIMD fit without spilling
    for (int i = tid; i < kNFft; i += 256) s_win[i] = window[i];
    for (int i = tid; i < 162; i += 256) s_sth[i] = i >= 2 ? tb.st320[i - 2] * 0.5f : (i == 0 ? 0.0f : -0.5f);
    __syncthreads(); // the only workgroup barrier: window and un-mixing table

    float* in = s_in[wave];
    float* tmag = s_mag[wave];
    float* tbin = s_bin[wave];
    // this lane's five float4 of the group starting at `base` (lanes past the end of the batch re-read the last
    // valid float4: the loads stay unconditional, their values are never used)
    auto fetch = [&](long base, f32x4 (&r)[5]) {
        const long left4 = (n_frames - base) * (kNFft / 4);
        const f32x4* src = reinterpret_cast<const f32x4*>(frames + base * kNFft);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            long i4 = lane + 64 * j;
            i4 = i4 < left4 ? i4 : left4 - 1;
            r[j] = src[i4];
        }
    };
    auto park = [&](const f32x4 (&r)[5]) {
#pragma unroll
        for (int j = 0; j < 5; ++j) reinterpret_cast<f32x4*>(in)[lane + 64 * j] = r[j];
    };
    const long stride = (long)gridDim.x * 4 * RB_WF;
    long base = ((long)blockIdx.x * 4 + wave) * RB_WF;
    f32x4 stage[5];
    if (base < n_frames) { fetch(base, stage); park(stage); }
    for (; base < n_frames; base += stride) {
        const bool more = base + stride < n_frames;
        if (more) fetch(base + stride, stage);
#pragma unroll 1
        for (int q = 0; q < RB_WF / 2; ++q) {
            __builtin_amdgcn_wave_barrier();
            const int fl = 2 * q + half; // frame of this half-wavefront within the group
            cpx v[5];
            {
                const float* x = in + fl * kNFft;
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const int n = 2 * (p + 32 * j);
                    const float2 xv = *reinterpret_cast<const float2*>(x + n);
                    const float2 wv = *reinterpret_cast<const float2*>(s_win + n);
                    v[j] = {xv.x * wv.x, xv.y * wv.y}; // loadSamplesFwd, FFT.zig:183-199
                }
            }
            wave_fft<5, 32, false>(v, tw, p);
            __builtin_amdgcn_wave_barrier();
            {
                float* z = in + fl * kNFft; // over the pair's own samples, which are in registers by now
#pragma unroll
                for (int k1 = 0; k1 < 5; ++k1) {
                    const int k = k1 + 5 * k2;
                    *reinterpret_cast<float2*>(z + 2 * k) = make_float2(v[k1].r, v[k1].i);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // un-mix: lanes 0..31 take the pair's first frame, lanes 32..63 the second; bins k = p, p + 32, p + 64
            // (<= 80) and their mirrors 160 - k.  Bin 80 is its own mirror: both forms are written, the X[ncfft - k]
            // one last, as kissfft does.
            if (base + fl < n_frames) {
                const float* z = in + fl * kNFft;
                float* mt = tmag + fl * kNBins;
                float* bt = tbin + (HAS_BINS ? fl * (kNBins * 2) : 0);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int k = p + 32 * u;
                    if (u < 2 || k <= 80) {
                        const int kn = 160 - k;
                        const int ksrc = (u == 0 && k == 0) ? 0 : kn; // z[160] does not exist: k = 0 pairs with itself
                        const float2 zk = *reinterpret_cast<const float2*>(z + 2 * k);
                        const float2 zn = *reinterpret_cast<const float2*>(z + 2 * ksrc);
                        const float2 st = *reinterpret_cast<const float2*>(s_sth + 2 * k);
                        cpx xk, xnk;
                        unmix_fwd_h({zk.x, zk.y}, {zn.x, zn.y}, {st.x, st.y}, xk, xnk);
                        if (HAS_BINS) {
                            *reinterpret_cast<float2*>(bt + 2 * k) = make_float2(xk.r, xk.i);
                            *reinterpret_cast<float2*>(bt + 2 * kn) = make_float2(xnk.r, xnk.i);
                        }
                        if (mag) { // v_sqrt_f32 (1 ulp): the batched magnitudes are a convenience output (FFT.zig:16-18)
                            mt[k] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.r, xk.r, xk.i * xk.i));
                            mt[kn] = __builtin_amdgcn_sqrtf(__builtin_fmaf(xnk.r, xnk.r, xnk.i * xnk.i));
                        }
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the next group's samples: the only wait for global memory in the loop, placed before this group's stores
        // are issued (vmcnt retires in order: a wait behind the stores would wait for them too)
        if (more) park(stage);
        __builtin_amdgcn_wave_barrier();
        const int nfr = (n_frames - base < RB_WF) ? (int)(n_frames - base) : RB_WF;
        auto flush = [&](float* dst, const float* tile, int nfl) {
            if (vec_ok) {
                for (int i = lane; i < nfl / 4; i += 64) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(tile)[i];
                for (int i = (nfl & ~3) + lane; i < nfl; i += 64) dst[i] = tile[i];
            } else {
                for (int i = lane; i < nfl; i += 64) dst[i] = tile[i];
            }
        };
        if (mag) flush(mag + base * kNBins, tmag, nfr * kNBins);
        if (HAS_BINS) flush(bins + base * (kNBins * 2), tbin, nfr * kNBins * 2);
    }
}

// ---- 320-point batch, eight frames per wavefront (large batches: fvad_launch_rfft_batch)
// A frame is EIGHT lanes with 20 complex points each, z[a + 8 j] in lane a:
//     Z[k1 + 20 k2] = sum_a W8^{a k2} ( W160^{a k1} sum_j z[a + 8 j] W20^{j k1} )
// a 20-point transform in registers (5 x dft4, twelve twiddles, 4 x dft5), one twiddle multiply, then three exchange stages
// across the frame's lanes, each ONE v_fmac_f32_dpp per scalar (lane ^ 7 = row_half_mirror with the rows renumbered as in
// vadfft1024_band_kernel, lane ^ 2, lane ^ 1) -- against five stages (one of them two permlane swaps) for 5 points x 32 lanes:
// ~90 VALU instructions per frame for the transform instead of ~113.  The last exchange stage has no twiddle and leaves MINUS
// the value on the odd rows, i.e. on bins 80..159: the un-mixing pass, which pairs every bin k <= 80 with 160 - k, reads them
// with the sign flipped (free: the negation folds into its adds).  Un-mixing through the wavefront's LDS slab as before, 8 x 81
// pairs over 64 lanes (92 % of the lanes busy instead of 84 %).
// Pipeline per wavefront: the next group's ten float4 per lane are fetched between this group's transform and its un-mixing pass
// (the transform's registers are free then) and parked in the slab after it; magnitudes and complex bins leave straight from the
// un-mixing pass as buffer stores (the group in the resource, constant lane offsets: no address arithmetic, no LDS tile -- 43 KB
// of LDS per workgroup, three workgroups per CU).
constexpr long kBatch8MinFrames = 16384;
constexpr int RB8_WAVES = 4;   // 4 x 10 KB slab + tables = 43 KB: three workgroups per CU
constexpr int RB8_FR = 8;      // frames per wavefront iteration
constexpr int RB8_FS = kNFft + 16; // floats per frame in the slab: 16 of padding, so that the 8-byte accesses of a half-wavefront (four
                                   // frames x eight lanes, 16 floats apart per frame step) cover all 64 banks instead of colliding 8 ways

// forward twiddle exp(-2 pi i m / 20); constants rounded from double
__device__ __forceinline__ cpx w20(int m)
{
    const float C[6] = {1.0f, 0.95105651629515357f, 0.80901699437494742f, 0.58778525229247313f, 0.30901699437494742f, 0.0f};
    m %= 20;
    const int quad = m / 5, r = m % 5;
    float c, sn;
    switch (quad) {
    case 0: c = C[r]; sn = C[5 - r]; break;
    case 1: c = -C[5 - r]; sn = C[r]; break;
    case 2: c = -C[r]; sn = -C[5 - r]; break;
    default: c = C[5 - r]; sn = -C[r]; break;
    }
    return {c, -sn};
}
// 20 points in registers: v[j] in; position 5 ka + kb holds X[ka + 4 kb] out (j = 5 j1 + j2:
// X[ka + 4 kb] = sum_j2 W5^{j2 kb} ( W20^{j2 ka} sum_j1 v[5 j1 + j2] W4^{j1 ka} ))
__device__ __forceinline__ void dft20(cpx (&v)[20])
{
#pragma unroll
    for (int j2 = 0; j2 < 5; ++j2) dft4<false>(v[j2], v[5 + j2], v[10 + j2], v[15 + j2]); // ka at position 5 ka + j2
#pragma unroll
    for (int ka = 1; ka < 4; ++ka)
#pragma unroll
        for (int j2 = 1; j2 < 5; ++j2) {
            const int m = j2 * ka;
            v[5 * ka + j2] = (m == 5) ? mul_mi<false>(v[5 * ka + j2]) : ((m == 10) ? cpx{-v[5 * ka + j2].r, -v[5 * ka + j2].i} : cmul_fma(v[5 * ka + j2], w20(m)));
        }
#pragma unroll
    for (int ka = 0; ka < 4; ++ka) {
        cpx t[5] = {v[5 * ka], v[5 * ka + 1], v[5 * ka + 2], v[5 * ka + 3], v[5 * ka + 4]};
        dft5<false>(t);
#pragma unroll
        for (int kb = 0; kb < 5; ++kb) v[5 * ka + kb] = t[kb];
    }
}

__global__ __launch_bounds__(64 * RB8_WAVES, 3) void rfft320_batch8_kernel(const float* __restrict__ frames, long n_frames,
                                                                         const float* __restrict__ window, FftTables tb,
                                                                         float* __restrict__ bins, float* __restrict__ mag)
{
    // dynamic LDS (so that the register allocator plans for three wavefronts per SIMD, which 43 KB per workgroup allow):
    // per wavefront a slab of 8 frames; the un-mixing table / 2, entry k for bin k (unmix_fwd_h); the window
    extern __shared__ __attribute__((aligned(16))) float rb8_smem[];
    float* s_sth = rb8_smem + RB8_WAVES * RB8_FR * RB8_FS;
    float* s_win = s_sth + 2 * 81 + 2;
    float* s_twl = s_win + kNFft; // [8 rows][20][2]: W160^{a k1} for the register at position 5 ka + kb (k1 = ka + 4 kb)
    float* s_ny = s_twl + 8 * 20 * 2;                                             // [wavefront][8][2]: X[160] (or |X[160]|) per frame
    unsigned short* s_map = reinterpret_cast<unsigned short*>(s_ny + RB8_WAVES * RB8_FR * 2); // [8 * 161]: where magnitude q of a group sits
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = lane >> 3; // frame of the group
    // row of the frame held by this lane: bits (l2, l1 ^ l2, l0 ^ l2), so that flipping bit 2 of `a` is lane ^ 7 (a DPP pattern)
    const int a = (lane & 4) | ((lane & 3) ^ ((lane & 4) ? 3 : 0));
    for (int i = tid; i < kNFft; i += 64 * RB8_WAVES) s_win[i] = window[i];
    for (int i = tid; i < 162; i += 64 * RB8_WAVES) s_sth[i] = i >= 2 ? tb.st320[i - 2] * 0.5f : (i == 0 ? 0.0f : -0.5f);
    for (int i = tid; i < 8 * 20; i += 64 * RB8_WAVES) {
        const int row = i / 20, r = i - 20 * row, k1 = r / 5 + 4 * (r % 5);
        s_twl[2 * i] = tb.tw160[2 * (row * k1)];
        s_twl[2 * i + 1] = tb.tw160[2 * (row * k1) + 1];
    }
    // magnitude q = 161 f + c of a group is parked in the slab slot its bin came from (float 320 f + 2 c; bin 160 has none: s_ny)
    for (int i = tid; i < RB8_FR * kNBins; i += 64 * RB8_WAVES) {
        const int fr = i / kNBins, c = i - kNBins * fr;
        s_map[i] = (unsigned short)(c < 160 ? RB8_FS * fr + 2 * c : 0x8000 + fr);
    }
    __syncthreads(); // the only workgroup barrier: window and un-mixing table

    // ---- per-lane constants: the stage twiddles (the 19 twiddles W160^{a k1} between the register transform and the exchange
    // stages are read from LDS where they are used: 38 registers less, which is what lets three wavefronts share a SIMD)
    const float* twl = s_twl + 40 * a;
    cpx tws[2]; // strides 4 and 2: 1 on the lower row, MINUS the twiddle on the upper (the butterfly forms mine - other there)
#pragma unroll
    for (int st = 0; st < 2; ++st) {
        const int h = 4 >> st;
        cpx t = {1.0f, 0.0f};
        if (a & h) {
            t = ld_tw(tb.tw160, (a & (h - 1)) * (80 / h)); // W_{2h}^{a mod h} = W160^{(a mod h) 80 / h}
            t = {-t.r, -t.i};
        }
        tws[st] = t;
    }
    const int k2 = (int)(__brev((unsigned)a) >> 29);

    float* in = rb8_smem + wave * (RB8_FR * RB8_FS);
    float* ny = s_ny + wave * (RB8_FR * 2);
    // outputs as buffer stores: the group's first frame in the resource (its size bounds the stores of a partial last group),
    // a lane's bins at constant offsets: frame f, bin i + 8 u and its mirror
    const int i8 = lane & 7;
    const unsigned vo_m = (unsigned)(f * kNBins + i8) * 4u, vo_mn = (unsigned)(f * kNBins + 160 - i8) * 4u;
    // this lane's ten float4 of the group starting at `base`: float4 (lane & 7) + 8 j of frame lane >> 3 (eight lanes cover 128
    // contiguous bytes of a frame per load), parked at the same place of the frame's padded slab row.  Frames past the end of
    // the batch re-read the batch's last frame: the loads stay unconditional, their values are never used
    auto fetch = [&](long base, f32x4 (&r)[10]) {
        long fr = base + f;
        fr = fr < n_frames ? fr : n_frames - 1;
        const f32x4* src = reinterpret_cast<const f32x4*>(frames + fr * kNFft) + (lane & 7);
#pragma unroll
        for (int j = 0; j < 10; ++j) r[j] = src[8 * j];
    };
    auto park = [&](const f32x4 (&r)[10]) {
        f32x4* dst = reinterpret_cast<f32x4*>(in + f * RB8_FS) + (lane & 7);
#pragma unroll
        for (int j = 0; j < 10; ++j) dst[8 * j] = r[j];
    };
    const long stride = (long)gridDim.x * RB8_WAVES * RB8_FR;
    long base = ((long)blockIdx.x * RB8_WAVES + wave) * RB8_FR;
    f32x4 stage[10];
    if (base < n_frames) { fetch(base, stage); park(stage); }
    for (; base < n_frames; base += stride) {
        const bool more = base + stride < n_frames;
        if (more) fetch(base + stride, stage); // a whole iteration to land
        __builtin_amdgcn_wave_barrier();
        float* z = in + f * RB8_FS;
        {
            cpx v[20];
#pragma unroll
            for (int j = 0; j < 20; ++j) {
                const int n = 2 * (a + 8 * j);
                const float2 xv = *reinterpret_cast<const float2*>(z + n);
                const float2 wv = *reinterpret_cast<const float2*>(s_win + n);
                v[j] = {xv.x * wv.x, xv.y * wv.y}; // loadSamplesFwd, FFT.zig:183-199
            }
            dft20(v);
#pragma unroll
            for (int r = 1; r < 20; ++r) {
                const float2 t = *reinterpret_cast<const float2*>(twl + 2 * r);
                v[r] = cmul_fma(v[r], {t.x, t.y});
            }
#pragma unroll
            for (int st = 0; st < 3; ++st) {
                const int h = 4 >> st;
                const float sgn = (a & h) ? -1.0f : 1.0f;
                dpp_block_begin();
#pragma unroll
                for (int r = 0; r < 20; ++r) {
                    dpp_butterfly_raw(v[r].r, sgn, h);
                    dpp_butterfly_raw(v[r].i, sgn, h);
                }
                dpp_block_end();
                if (h > 1) {
#pragma unroll
                    for (int r = 0; r < 20; ++r) v[r] = cmul_fma(v[r], tws[st]);
                }
            }
            __builtin_amdgcn_wave_barrier();
            // over the frame's own samples, which are in registers by now; rows with a odd (bins 80..159) hold MINUS the bin
#pragma unroll
            for (int r = 0; r < 20; ++r) {
                const int k = (r / 5 + 4 * (r % 5)) + 20 * k2;
                *reinterpret_cast<float2*>(z + 2 * k) = make_float2(v[r].r, v[r].i);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // un-mix: lane (f, i) takes bins k = i + 8 u <= 80 of frame f and their mirrors 160 - k.  Bin 80 is its own mirror:
        // both forms are written, the X[ncfft - k] one last, as kissfft does.
        {
            const int nfr = (n_frames - base < RB8_FR) ? (int)(n_frames - base) : RB8_FR;
            const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(mag ? mag + base * kNBins : nullptr, 0, mag ? nfr * kNBins * 4 : 0, 0x00020000);
            // u = 0 (bin 0 pairs with itself and is not negated) and u = 10 (only bin 80, its own mirror) are written out; the
            // nine iterations between them are ONE rolled loop: unrolled, the compiler hoists all 33 LDS reads to the top and the
            // register count no longer leaves room for the staged group
            auto pair = [&](int u, bool first, bool last) {
                const int k = i8 + 8 * u;
                if (!last || k <= 80) {
                    const int kn = 160 - k;
                    const bool dc = first && k == 0;
                    const int ksrc = dc ? 0 : kn; // z[160] does not exist: k = 0 pairs with itself
                    float2 zk = *reinterpret_cast<const float2*>(z + 2 * k);
                    float2 zn = *reinterpret_cast<const float2*>(z + 2 * ksrc);
                    if (last) zk = make_float2(-zk.x, -zk.y);              // k = 80: stored negated
                    if (first) { const float sg = dc ? 1.0f : -1.0f; zn = make_float2(zn.x * sg, zn.y * sg); } // bin 0 is not
                    else zn = make_float2(-zn.x, -zn.y);                    // bins 80..159 are stored negated
                    const float2 st = *reinterpret_cast<const float2*>(s_sth + 2 * k);
                    cpx xk, xnk;
                    unmix_fwd_h({zk.x, zk.y}, {zn.x, zn.y}, {st.x, st.y}, xk, xnk);
                    // Outputs are parked in the slab slots the pair's own bins came from (no other lane reads them) and flushed
                    // below as contiguous stores; bin 160 (the mirror of bin 0) has no slot in a 160-bin slab.  Complex bins
                    // take the slots when they are asked for -- the magnitudes of such a call leave from here, 32 bytes per
                    // frame and store -- the magnitudes otherwise.
                    // v_sqrt_f32 (1 ulp): the batched magnitudes are a convenience output (FFT.zig:16-18)
                    const float mk = __builtin_amdgcn_sqrtf(__builtin_fmaf(xk.r, xk.r, xk.i * xk.i));
                    const float mn = __builtin_amdgcn_sqrtf(__builtin_fmaf(xnk.r, xnk.r, xnk.i * xnk.i));
                    if (bins) {
                        *reinterpret_cast<float2*>(z + 2 * k) = make_float2(xk.r, xk.i);
                        if (dc) *reinterpret_cast<float2*>(ny + 2 * f) = make_float2(xnk.r, xnk.i);
                        else *reinterpret_cast<float2*>(z + 2 * kn) = make_float2(xnk.r, xnk.i);
                        if (mag) {
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mk), rs_m, vo_m + 32u * u, 0, 0);
                            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, mn), rs_m, vo_mn - 32u * u, 0, 0);
                        }
                    } else {
                        z[2 * k] = mk;
                        if (dc) ny[2 * f] = mn; else z[2 * kn] = mn;
                    }
                }
            };
            pair(0, true, false);
#pragma unroll 3
            for (int u = 1; u < 10; ++u) pair(u, false, false);
            pair(10, false, true);
        }
        __builtin_amdgcn_wave_barrier();
        {   // 8 x 161 parked values as 21 contiguous stores (the resource's size cuts a partial last group)
            const int nfr = (n_frames - base < RB8_FR) ? (int)(n_frames - base) : RB8_FR;
            if (bins) {
                const auto rs_b = __builtin_amdgcn_make_buffer_rsrc(bins + base * (kNBins * 2), 0, nfr * kNBins * 8, 0x00020000);
#pragma unroll 3
                for (int t = 0; t < 21; ++t) {
                    const int q = lane + 64 * t;
                    if (t < 20 || q < RB8_FR * kNBins) {
                        const unsigned m = s_map[q];
                        const float2 val = *reinterpret_cast<const float2*>((m & 0x8000u) ? ny + 2 * (m & 7u) : in + m);
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, (f32x2v){val.x, val.y}), rs_b, (unsigned)q * 8u, 0, 0);
                    }
                }
            } else if (mag) {
                const auto rs_m = __builtin_amdgcn_make_buffer_rsrc(mag + base * kNBins, 0, nfr * kNBins * 4, 0x00020000);
#pragma unroll 3
                for (int t = 0; t < 21; ++t) {
                    const int q = lane + 64 * t;
                    if (t < 20 || q < RB8_FR * kNBins) {
                        const unsigned m = s_map[q];
                        const float val = (m & 0x8000u) ? ny[2 * (m & 7u)] : in[m];
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, val), rs_m, (unsigned)q * 4u, 0, 0);
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // the next group's samples: the only wait for global memory in the loop, placed before this group's stores
        // are issued (vmcnt retires in order: a wait behind the stores would wait for them too)
        if (more) park(stage);
    }
}

void fvad_launch_rfft_batch(const float* frames, long n_frames, int n_fft, const float* window,
                            FftTables tb, VadFftPlan pl, float* bins_or_null, float* mag_or_null,
                            hipStream_t stream)
{
    if (n_frames <= 0) return;
    if (n_fft != kNFft && pl.generic) {
        launch_rfft_generic(frames, n_frames, window, pl, bins_or_null, mag_or_null, nullptr, 0, 0, 0, 0, stream);
    } else if (n_fft != kNFft) {
        VADFFT_DISPATCH(n_fft, hipLaunchKernelGGL(rfftN_batch_kernel<R>, dim3((unsigned)((n_frames + 3) / 4)), dim3(256), 0,
                                                  stream, frames, n_frames, window, pl, bins_or_null, mag_or_null))
    } else {
        // From 16384 frames on (two thirds of one resident round of it) the eight-frames-per-wavefront kernel: 6-8 % faster at
        // 2^20 frames; below, the four-frame kernel, whose single iteration is 2.6 us shorter (1024 frames: 5.7 against 8.3 us).
        // The two factor the transform differently and agree to rounding (~1e-7 of a frame's largest bin), not bit for bit.
        if (n_frames >= kBatch8MinFrames) {
            long groups8 = (n_frames + RB8_WAVES * RB8_FR - 1) / (RB8_WAVES * RB8_FR); // 4 wavefronts x 8 frames per workgroup and iteration
            // persistent beyond one resident round (three workgroups per CU): no tail of a partial last round
            static int n_cu_cached = 0;
            if (!n_cu_cached) {
                int dev = 0, cu = 0;
                if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cu > 0) n_cu_cached = cu;
                else n_cu_cached = 256;
            }
            if (groups8 > 3L * n_cu_cached) groups8 = 3L * n_cu_cached;
            constexpr size_t lds8 = (size_t)(RB8_WAVES * RB8_FR * RB8_FS + 2 * 81 + 2 + kNFft + 8 * 20 * 2 + RB8_WAVES * RB8_FR * 2) * sizeof(float) + RB8_FR * kNBins * 2;
            hipLaunchKernelGGL(rfft320_batch8_kernel, dim3((unsigned)groups8), dim3(64 * RB8_WAVES), lds8, stream, frames, n_frames, window, tb,
                               bins_or_null, mag_or_null);
            return;
        }
        long groups = (n_frames + 15) / 16; // 4 wavefronts x 4 frames per workgroup and iteration
        if (groups > 2048) groups = 2048;
        // flat float4 stores need 16-byte aligned outputs (a group's tile starts at a multiple of 8 x 161 floats)
        const int vec_ok = (((uintptr_t)bins_or_null | (uintptr_t)mag_or_null) % 16) == 0;
        if (bins_or_null)
            hipLaunchKernelGGL(rfft320_batch_kernel<true>, dim3((unsigned)groups), dim3(256), 0, stream, frames,
                               n_frames, window, tb, bins_or_null, mag_or_null, vec_ok);
        else
            hipLaunchKernelGGL(rfft320_batch_kernel<false>, dim3((unsigned)groups), dim3(256), 0, stream, frames,
                               n_frames, window, tb, bins_or_null, mag_or_null, vec_ok);
    }
}

// FFT.invFft for many 320-point frames: bins [n][161][2] -> out [n][320], unscaled
__global__ __launch_bounds__(256) void irfft320_batch_kernel(const float* __restrict__ bins,
                                                             long n_frames, FftTables tb,
                                                             float* __restrict__ out)
{
    __shared__ __attribute__((aligned(16))) float zb[4][2][2 * 160];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int half = lane >> 5;
    const int p = lane & 31;
    LaneTw<5, 32> tw;
    lane_tw_load<5, 32, true>(tw, tb.tw160, p);
    const int k2 = bitrev_lane<32>(p);
    for (long base = (long)blockIdx.x * 8; base < n_frames; base += (long)gridDim.x * 8) {
        for (int item = lane; item < 2 * 81; item += 64) {
            const int hh = item / 81;
            const int k = item - hh * 81;
            const long fr = base + 2 * wave + hh;
            if (fr < n_frames) {
                const float* b = bins + fr * kNBins * 2;
                const int kn = 160 - k;
                const cpx yk = {b[2 * k], b[2 * k + 1]};
                const cpx ynk = {b[2 * kn], b[2 * kn + 1]};
                float* z = zb[wave][hh];
                if (k == 0) {
                    z[0] = yk.r + ynk.r;
                    z[1] = yk.r - ynk.r;
                } else {
                    cpx tk, tnk;
                    premix_inv(yk, ynk, {tb.st320[2 * (k - 1)], -tb.st320[2 * (k - 1) + 1]}, tk, tnk);
                    z[2 * k] = tk.r; z[2 * k + 1] = tk.i;
                    z[2 * kn] = tnk.r; z[2 * kn + 1] = tnk.i;
                }
            }
        }
        __syncthreads();
        const long frame = base + 2 * wave + half;
        if (frame < n_frames) {
            const float* z = zb[wave][half];
            cpx v[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int n = p + 32 * j;
                v[j] = {z[2 * n], z[2 * n + 1]};
            }
            wave_fft<5, 32, true>(v, tw, p);
            float* o = out + frame * kNFft;
#pragma unroll
            for (int k1 = 0; k1 < 5; ++k1) {
                const int n = 2 * (k1 + 5 * k2);
                *reinterpret_cast<float2*>(o + n) = make_float2(v[k1].r, v[k1].i);
            }
        }
        __syncthreads();
    }
}

void fvad_launch_irfft_batch(const float* bins, long n_frames, FftTables tb, float* out,
                             hipStream_t stream)
{
    if (n_frames <= 0) return;
    long groups = (n_frames + 7) / 8;
    if (groups > 4096) groups = 4096;
    hipLaunchKernelGGL(irfft320_batch_kernel, dim3((unsigned)groups), dim3(256), 0, stream, bins,
                       n_frames, tb, out);
}
