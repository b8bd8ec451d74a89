// kernels_vadfft.hip -- K4: periodic-Hann real FFT of the VAD side, |X| * norm, band sum (BufferedFFT.zig:162-202).
//   vadfft_kernel<R> / vadfft_jobs_kernel<R>   one wavefront per frame of 512 / 1024 / 2048 samples, every bin (the magnitude tap)
//   vadfft1024_band_kernel                     1024 points, band sum only: four frames per wavefront, pruned
// (the shared wavefront FFT scheme and rfftN_wave / rfftN_bin: fft_device.h)
#include "fft_device.h"

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

int fvad_launch_vadfft_jobs(const VadFftJob* jobs, int n_jobs, long max_frames, VadFftPlan pl,
                            int min_bin, int max_bin, hipStream_t stream, int any_bins, int n_cu, int plain)
{
    if (n_jobs <= 0 || max_frames <= 0) return (int)hipSuccess;
    if (pl.generic)
        return fvad_launch_rfft_generic_any(nullptr, 0, nullptr, pl, nullptr, nullptr, jobs, n_jobs, max_frames, min_bin, max_bin, stream);
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
            const hipError_t e = hipFuncSetAttribute((const void*)vadfft1024_band_kernel<11, 43>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e; // no kernel has written the band sums: the caller must not return FVAD_OK
            hipLaunchKernelGGL((vadfft1024_band_kernel<11, 43>), dim3((unsigned)per_job, (unsigned)n_jobs), dim3(256), lds, stream, jobs, pl, min_bin, max_bin, plain);
        } else {
            const hipError_t e = hipFuncSetAttribute((const void*)vadfft1024_band_kernel<0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return (int)e;
            hipLaunchKernelGGL((vadfft1024_band_kernel<0, 0>), dim3((unsigned)per_job, (unsigned)n_jobs), dim3(256), lds, stream, jobs, pl, min_bin, max_bin, plain);
        }
    }
    return (int)hipGetLastError();
}

