// Calibration of rocprofv3's FETCH_SIZE on gfx950 for the two read shapes of this library (run under
// `rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- ./fetch_calib`): every kernel reads a known
// number of bytes exactly once from a 2 GiB buffer (far beyond the 256 MiB Infinity Cache).
//   fetch_stream   : 16 B per lane, 1 KB contiguous per wave-instruction (K1 / K3 / K4, LDS-DMA weight slabs)
//   fetch_rows<S>  : the MFMA operand layout -- lane (m = l & 15, q = l >> 4) reads the float4 at row m, bytes
//                    16 q of a 64-byte segment: 16 rows x 64 B per wave-instruction, rows S bytes apart
//                    (S = 1600: h rows of gru_rec3, S = 704: feature rows, S = 4800: gi rows), super-step after
//                    super-step until the row is consumed -- what panel_gemm3 / gru_rec3 do with activations
// The guide's rule (MI355X_MICROARCH.md, HBM): FETCH_SIZE reads 1/2 of a wide coalesced stream.  Print-out:
// bytes each kernel actually read; compare with FETCH_SIZE * 1024 from the profiler (tools/pmc_kernel.py).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fetch_stream(const f32x4* __restrict__ src, size_t n4, float* sink)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) acc += src[i];
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int STRIDE_BYTES>
__global__ __launch_bounds__(256) void fetch_rows(const float* __restrict__ src, size_t n_rows, float* sink)
{
    constexpr int ROW_F = STRIDE_BYTES / 4;        // floats per row
    constexpr int S_STEPS = STRIDE_BYTES / 64;     // 64-byte segments per row
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    // one wavefront owns 16 consecutive rows and walks them super-step by super-step
    for (size_t tile = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); tile * 16 + 16 <= n_rows; tile += (size_t)gridDim.x * 4) {
        const float* row = src + (tile * 16 + m) * ROW_F + 4 * q;
#pragma unroll 5
        for (int S = 0; S < S_STEPS; ++S) acc += *reinterpret_cast<const f32x4*>(row + 16 * S);
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

// gru_rec3_kernel's own reads of gi (kernels_nn.hip: gi_off = (m T 1200 + 4 q) 4 bytes, three float4 loads per unit tile at
// +192 J, +64, +128): a wavefront owns 16 SEQUENCES, so the 16 rows of one wave-instruction are T x 4800 bytes apart, and a
// row's 4800 bytes are consumed as 25 chunks of 192 contiguous bytes, one time step after the other.  12 wavefronts per
// workgroup like the kernel's <12, 2> instance.  Every byte of [n_seq][T][1200] f32 is read exactly once.
template <int T>
__global__ __launch_bounds__(768) void fetch_gi_tiles(const float* __restrict__ gi, size_t n_seq, float* sink)
{
    const int lane = threadIdx.x & 63, m = lane & 15, q = lane >> 4, w = threadIdx.x >> 6;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (size_t seq0 = ((size_t)blockIdx.x * 12 + w) * 16; seq0 + 16 <= n_seq; seq0 += (size_t)gridDim.x * 192) {
        const float* base = gi + (seq0 + m) * (size_t)T * 1200 + 4 * q;
        for (int t = 0; t < T; ++t)
#pragma unroll 5
            for (int J = 0; J < 25; ++J) {
                const float* p = base + (size_t)t * 1200 + 48 * J;
                acc += *reinterpret_cast<const f32x4*>(p);
                acc += *reinterpret_cast<const f32x4*>(p + 16);
                acc += *reinterpret_cast<const f32x4*>(p + 32);
            }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

int main()
{
    const size_t bytes = 2ull << 30;
    float *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(buf, 0, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(fetch_stream, dim3(4096), dim3(256), 0, 0, (const f32x4*)buf, bytes / 16, sink);
        hipLaunchKernelGGL(fetch_rows<1600>, dim3(4096), dim3(256), 0, 0, buf, bytes / 1600 / 16 * 16, sink);
        hipLaunchKernelGGL(fetch_rows<704>, dim3(4096), dim3(256), 0, 0, buf, bytes / 704 / 16 * 16, sink);
        hipLaunchKernelGGL(fetch_rows<4800>, dim3(4096), dim3(256), 0, 0, buf, bytes / 4800 / 16 * 16, sink);
        hipLaunchKernelGGL(fetch_gi_tiles<54>, dim3(256), dim3(768), 0, 0, buf, bytes / (54 * 4800) / 16 * 16, sink);
    }
    hipDeviceSynchronize();
    printf("fetch_stream: %zu bytes\nfetch_rows<1600>: %zu bytes\nfetch_rows<704>: %zu bytes\nfetch_rows<4800>: %zu bytes\nfetch_gi_tiles<54>: %zu bytes\n", bytes,
           bytes / 1600 / 16 * 16 * 1600, bytes / 704 / 16 * 16 * 704, bytes / 4800 / 16 * 16 * 4800, bytes / (54 * 4800) / 16 * 16 * (size_t)(54 * 4800));
    return 0;
}
