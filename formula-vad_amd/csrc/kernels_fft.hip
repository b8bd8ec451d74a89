// kernels_fft.hip -- FFT.fft / FFT.invFft for many frames (B3; BASELINE config 2): the 320-point batch in two forms (four and
// eight frames per wavefront), the N-point batch (512 / 1024 / 2048), the inverse 320-point batch.
// (the shared wavefront FFT scheme: fft_device.h; K1 / K3: kernels_stft.hip; K4: kernels_vadfft.hip; other sizes: kernels_fftgen.hip)
#include "fft_device.h"

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

int fvad_launch_rfft_batch(const float* frames, long n_frames, int n_fft, const float* window,
                           FftTables tb, VadFftPlan pl, float* bins_or_null, float* mag_or_null,
                           hipStream_t stream)
{
    if (n_frames <= 0) return (int)hipSuccess;
    if (n_fft != kNFft && pl.generic) {
        return fvad_launch_rfft_generic_any(frames, n_frames, window, pl, bins_or_null, mag_or_null, nullptr, 0, 0, 0, 0, stream);
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
            return (int)hipGetLastError();
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
    return (int)hipGetLastError();
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
