// kernels_stft.hip -- the NSNet2 side of the spectral front end on gfx950.
//
//   K1 stft_kernel    chunk RMS (BufferedVolumeAnalyzer.zig:48-69, audio_utils.zig:14-24),
//                     /3 decimation (resample.zig:9-29), sqrt-Hann 320-point real FFT
//                     (NSNet2.zig:239-264 -> FFT.zig:85-113), log-power features (NSNet2.zig:266-287)
//   K3 istft_kernel   gain (NSNet2.zig:289-310), inverse real FFT + window + overlap-add
//                     (NSNet2.zig:312-339), x3 linear upsample (resample.zig:32-79)
// (the shared wavefront FFT scheme: fft_device.h)
#include "fft_device.h"

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

