// small_gemm.hip -- the small-batch (64-row panel) GEMM of kernels_nn.hip in isolation: the shipped shape (one 16-deep
// super-step per barrier, NT tiles per workgroup) against deeper phases (KS super-steps per barrier) and narrower
// column blocks, on the four layer shapes of BASELINE config 3's 82-chunk batch (96 padded sequences) and of a
// one-chunk push.  Every variant keeps each output's k-ordered fma chain, so all of them must agree bit for bit.
//   ./small_gemm            (gfx950 only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int NT, int KS>
__global__ __launch_bounds__(256) void gemm_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wfrag,
                                                   const float* __restrict__ bias, float* __restrict__ C, int ldc, int S_steps,
                                                   int n_valid_tiles)
{
    __shared__ __attribute__((aligned(16))) float slab[2][KS * NT * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int nblk = blockIdx.y;
    const unsigned row = (blockIdx.x * 4 + wave) * 16 + m;
    const float* a_ptr = A + (size_t)row * (size_t)lda + 4 * q;
    const float* w_src = Wfrag + (size_t)nblk * S_steps * (NT * 256);
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int SLAB_F4 = KS * NT * 64;
    constexpr int PER_T = (SLAB_F4 + 255) / 256;
    const int n_ph = (S_steps + KS - 1) / KS;
    f32x4 a_cur[KS];
    {
        const int ks0 = S_steps < KS ? S_steps : KS;
        const f32x4* src = reinterpret_cast<const f32x4*>(w_src);
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * 256;
            if (idx < ks0 * NT * 64) reinterpret_cast<f32x4*>(slab[0])[idx] = src[idx];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s)
            if (s < ks0) a_cur[s] = *reinterpret_cast<const f32x4*>(a_ptr + 16 * s);
    }
    __syncthreads();
    for (int p = 0; p < n_ph; ++p) {
        const int cur = p & 1;
        const int left = S_steps - p * KS;
        const int ks_cur = left < KS ? left : KS;
        const int ks_next = (left - KS) < KS ? (left - KS) : KS; // <= 0: none
        f32x4 stage[PER_T];
        f32x4 a_next[KS];
        {
            const f32x4* src = reinterpret_cast<const f32x4*>(w_src + (size_t)(p + 1) * KS * (NT * 256));
#pragma unroll
            for (int i = 0; i < PER_T; ++i) {
                const int idx = tid + i * 256;
                if (idx < ks_next * NT * 64) stage[i] = src[idx];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s)
                if (s < ks_next) a_next[s] = *reinterpret_cast<const f32x4*>(a_ptr + 16 * ((p + 1) * KS + s));
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            if (s < ks_cur) {
                const f32x4* wl = reinterpret_cast<const f32x4*>(slab[cur]) + s * NT * 64 + lane;
                f32x4 w4[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t) w4[t] = wl[t * 64];
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].x, a_cur[s].x, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].y, a_cur[s].y, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].z, a_cur[s].z, acc[t]);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = MFMA16(w4[t].w, a_cur[s].w, acc[t]);
            }
        }
#pragma unroll
        for (int i = 0; i < PER_T; ++i) {
            const int idx = tid + i * 256;
            if (idx < ks_next * NT * 64) reinterpret_cast<f32x4*>(slab[cur ^ 1])[idx] = stage[i];
        }
#pragma unroll
        for (int s = 0; s < KS; ++s) a_cur[s] = a_next[s];
        __syncthreads();
    }
    float* c_ptr = C + (size_t)row * (size_t)ldc + nblk * (NT * 16) + 4 * q;
    const float* b_ptr = bias + nblk * (NT * 16) + 4 * q;
    const int valid_t = n_valid_tiles - nblk * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < valid_t) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b_ptr + 16 * t);
            f32x4 v = acc[t] + b4;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4*>(c_ptr + 16 * t) = v;
        }
    }
}


// v2: the reduction length S (super-steps) is a template parameter and the phase loop is fully unrolled, so that no
// load sits under a run-time predicate (the compiler put s_waitcnt vmcnt(0) in the middle of the MFMA section of the
// predicated form above); PF = how many phases ahead the slab and activation loads run (LDS holds PF + 1 slabs).
template <int NT, int KS, int S, int PF>
__global__ __launch_bounds__(256) void gemm2_kernel(const float* __restrict__ A, int lda, const float* __restrict__ Wfrag,
                                                    const float* __restrict__ bias, float* __restrict__ C, int ldc, int n_valid_tiles)
{
    constexpr int NB = PF + 1;
    __shared__ __attribute__((aligned(16))) float slab[NB][KS * NT * 256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 15, q = lane >> 4;
    const int nblk = blockIdx.y;
    const unsigned row = (blockIdx.x * 4 + wave) * 16 + m;
    const float* a_ptr = A + (size_t)row * (size_t)lda + 4 * q;
    const f32x4* w_src = reinterpret_cast<const f32x4*>(Wfrag + (size_t)nblk * S * (NT * 256));
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr int NP = (S + KS - 1) / KS;
    constexpr int PER_T = (KS * NT * 64 + 255) / 256;
    f32x4 stage[NB][PER_T];
    f32x4 a_q[NB][KS];
    auto ks_of = [](int p) { return (S - p * KS) < KS ? (S - p * KS) : KS; };
    auto load = [&](int p) { // phase p's slab -> stage registers, activations -> a_q
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
            to_lds(p + 1); // into the buffer last read in phase p + 1 - NB <= p - 1: every wavefront is past that barrier
            __syncthreads();
        }
    }
    float* c_ptr = C + (size_t)row * (size_t)ldc + nblk * (NT * 16) + 4 * q;
    const float* b_ptr = bias + nblk * (NT * 16) + 4 * q;
    const int valid_t = n_valid_tiles - nblk * NT;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        if (t < valid_t) {
            const f32x4 b4 = *reinterpret_cast<const f32x4*>(b_ptr + 16 * t);
            f32x4 v = acc[t] + b4;
            v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
            *reinterpret_cast<f32x4*>(c_ptr + 16 * t) = v;
        }
    }
}

// host: W[n][k] row-major -> fragment blocks [n_blocks][S][NT][64][4]
static std::vector<float> pack(const std::vector<float>& W, int N, int K, int NT, int S)
{
    const int n_tiles = (N + 15) / 16, n_blocks = (n_tiles + NT - 1) / NT;
    std::vector<float> out((size_t)n_blocks * S * NT * 256, 0.f);
    for (int b = 0; b < n_blocks; ++b)
        for (int s = 0; s < S; ++s)
            for (int t = 0; t < NT; ++t)
                for (int l = 0; l < 64; ++l)
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * (b * NT + t) + (l & 15), k = 16 * s + 4 * (l >> 4) + r;
                        if (n < N && k < K) out[(((size_t)(b * S + s) * NT + t) * 64 + l) * 4 + r] = W[(size_t)n * K + k];
                    }
    return out;
}

struct Shape { const char* name; int rows, K, N; };

template <int NT, int KS>
static double run(const Shape& sh, const float* dA, int lda, const std::vector<float>& W, const float* dbias, float* dC, int ldc,
                  std::vector<float>* result)
{
    const int S = (sh.K + 15) / 16, n_tiles = (sh.N + 15) / 16, n_blocks = (n_tiles + NT - 1) / NT;
    std::vector<float> wf = pack(W, sh.N, sh.K, NT, S);
    float* dW; CK(hipMalloc(&dW, wf.size() * 4)); CK(hipMemcpy(dW, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, (size_t)sh.rows * ldc * 4));
    dim3 grid(sh.rows / 64, n_blocks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gemm_kernel<NT, KS>), grid, dim3(256), 0, 0, dA, lda, dW, dbias, dC, ldc, S, n_tiles);
    CK(hipEventRecord(e0));
    const int reps = 200;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm_kernel<NT, KS>), grid, dim3(256), 0, 0, dA, lda, dW, dbias, dC, ldc, S, n_tiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (result) { result->resize((size_t)sh.rows * ldc); CK(hipMemcpy(result->data(), dC, result->size() * 4, hipMemcpyDeviceToHost)); }
    CK(hipFree(dW));
    return ms * 1e3 / reps;
}


template <int NT, int KS, int S, int PF>
static double run2(const Shape& sh, const float* dA, int lda, const std::vector<float>& W, const float* dbias, float* dC, int ldc,
                   std::vector<float>* result)
{
    if ((sh.K + 15) / 16 != S) return -1.0;
    const int n_tiles = (sh.N + 15) / 16, n_blocks = (n_tiles + NT - 1) / NT;
    std::vector<float> wf = pack(W, sh.N, sh.K, NT, S);
    float* dW; CK(hipMalloc(&dW, wf.size() * 4)); CK(hipMemcpy(dW, wf.data(), wf.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, (size_t)sh.rows * ldc * 4));
    dim3 grid(sh.rows / 64, n_blocks);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((gemm2_kernel<NT, KS, S, PF>), grid, dim3(256), 0, 0, dA, lda, dW, dbias, dC, ldc, n_tiles);
    CK(hipEventRecord(e0));
    const int reps = 200;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((gemm2_kernel<NT, KS, S, PF>), grid, dim3(256), 0, 0, dA, lda, dW, dbias, dC, ldc, n_tiles);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (result) { result->resize((size_t)sh.rows * ldc); CK(hipMemcpy(result->data(), dC, result->size() * 4, hipMemcpyDeviceToHost)); }
    CK(hipFree(dW));
    return ms * 1e3 / reps;
}

int main()
{
    const Shape shapes[] = {
        {"gi1f 5184x176->1200", 5184, 176, 1200}, {"fc2 4800x400->600", 4800, 400, 600}, {"fc3 4800x600->600", 4800, 600, 600},
        {"fc4 4800x600->161", 4800, 600, 161},    {"gi1f 64x176->1200", 64, 176, 1200},   {"fc3 64x600->600", 64, 600, 600},
        {"fc4 64x600->161", 64, 600, 161},        {"gi1f 1088x176->1200", 1088, 176, 1200}, {"fc3 1024x600->600", 1024, 600, 600},
        {"fc4 1024x600->161", 1024, 600, 161},    {"fc3 16384x600->600", 16384, 600, 600},
        // the 82-chunk batch's REAL rows: 4100 = 64 full panels + 4 rows (65 panels as launched; 64 if the remainder went elsewhere)
        {"fc2 4160x400->600", 4160, 400, 600},    {"fc2 4096x400->600", 4096, 400, 600},  {"fc3 4160x600->600", 4160, 600, 600},
        {"fc3 4096x600->600", 4096, 600, 600},    {"fc4 4160x600->161", 4160, 600, 161},  {"fc4 4096x600->161", 4096, 600, 161},
    };
    for (const Shape& sh : shapes) {
        const int lda = 640 > sh.K ? 640 : sh.K, ldc = 1280;
        std::vector<float> A((size_t)sh.rows * lda), W((size_t)sh.N * sh.K), bias(1280);
        srand(1);
        for (auto& x : A) x = (float)rand() / RAND_MAX - 0.5f;
        for (auto& x : W) x = ((float)rand() / RAND_MAX - 0.5f) * 0.1f;
        for (auto& x : bias) x = (float)rand() / RAND_MAX - 0.3f;
        float *dA, *dB, *dC;
        CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, bias.size() * 4)); CK(hipMalloc(&dC, (size_t)sh.rows * ldc * 4));
        CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
        std::vector<float> ref, got;
        printf("%s\n", sh.name);
        auto report = [&](const char* tag, double us, bool check) {
            bool same = true;
            if (check) {
                for (int r = 0; r < sh.rows && same; ++r)
                    if (memcmp(&ref[(size_t)r * ldc], &got[(size_t)r * ldc], sizeof(float) * sh.N)) same = false;
            }
            printf("  %-18s %7.2f us  %s\n", tag, us, check ? (same ? "same bits" : "DIFFERENT") : "(reference)");
        };
        double us;
#define RUN(NT, KS) us = run<NT, KS>(sh, dA, lda, W, dB, dC, ldc, &got); report("NT" #NT " KS" #KS, us, true);
        us = run<5, 1>(sh, dA, lda, W, dB, dC, ldc, &ref); report("NT5 KS1", us, false);
#define RUN2(NT, KS, S, PF) us = run2<NT, KS, S, PF>(sh, dA, lda, W, dB, dC, ldc, &got); if (us >= 0) report("v2 NT" #NT " KS" #KS " PF" #PF, us, true);
#define ALLS(NT, KS, PF) RUN2(NT, KS, 11, PF) RUN2(NT, KS, 25, PF) RUN2(NT, KS, 38, PF)
        ALLS(4, 2, 2) ALLS(4, 2, 3) ALLS(4, 2, 4) ALLS(4, 1, 3) ALLS(4, 1, 4) ALLS(4, 1, 6) ALLS(3, 2, 3) ALLS(3, 1, 4) ALLS(2, 2, 3) ALLS(2, 2, 4) ALLS(2, 1, 6) ALLS(2, 4, 2)
        ALLS(1, 4, 2) ALLS(1, 2, 4) ALLS(5, 2, 3) ALLS(6, 2, 2) ALLS(5, 1, 4) ALLS(5, 1, 3) ALLS(3, 1, 6) ALLS(6, 1, 3) ALLS(8, 1, 2)
        CK(hipFree(dA)); CK(hipFree(dB)); CK(hipFree(dC));
    }
    return 0;
}
