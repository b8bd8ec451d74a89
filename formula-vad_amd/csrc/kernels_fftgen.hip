// kernels_fftgen.hip -- real FFTs of any even size (FFT.init takes whatever kissfft factors, FFT.zig:35-60): boundary
// completeness, not a hot path.
#include "fft_device.h"

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

int fvad_launch_irfft_generic(const float* bins, long n_frames, VadFftPlan pl, float* out, hipStream_t stream)
{
    if (n_frames <= 0) return (int)hipSuccess;
    const size_t lds = (size_t)(pl.n / 2 + 1) * 2 * sizeof(cpx); // up to 131 KB: the attribute call can fail
    const hipError_t e = hipFuncSetAttribute((const void*)irfft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(irfft_generic_kernel, dim3((unsigned)n_frames), dim3(256), lds, stream, bins, n_frames, pl, out);
    return (int)hipGetLastError();
}

int fvad_launch_rfft_generic_any(const float* frames, long n_frames, const float* window, VadFftPlan pl, float* bins, float* mag,
                                 const VadFftJob* jobs, int n_jobs, long max_frames, int min_bin, int max_bin, hipStream_t stream)
{
    const size_t lds = (size_t)(pl.n / 2 + 1) * 2 * sizeof(cpx);
    const hipError_t e = hipFuncSetAttribute((const void*)rfft_generic_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return (int)e;
    const dim3 grid((unsigned)(jobs ? max_frames : n_frames), (unsigned)(jobs ? n_jobs : 1));
    hipLaunchKernelGGL(rfft_generic_kernel, grid, dim3(256), lds, stream, frames, n_frames, window, pl, bins, mag, jobs, min_bin, max_bin);
    return (int)hipGetLastError();
}
