// log10_pos (csrc/device_math.h) against the device library's log10f, bit for bit, over EVERY float from 1e-12 (the floor K1
// applies to a bin's power, NSNet2.zig:275) up to +inf.  make -C tools log10_check && tools/log10_check
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>
#include "../formula-vad_amd/csrc/device_math.h"

__global__ void check(uint32_t lo, uint32_t hi, unsigned long long* bad, uint32_t* first_bad)
{
    const unsigned long long n = (unsigned long long)hi - lo + 1;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint32_t bits = lo + (uint32_t)i;
        const float x = __uint_as_float(bits);
        const float a = log10f(x), b = log10_pos(x);
        if (__float_as_uint(a) != __float_as_uint(b)) {
            if (atomicAdd(bad, 1ull) == 0) *first_bad = bits;
        }
    }
}

int main()
{
    const float floor_v = 1.0f / 1e12f;
    uint32_t lo, hi = 0x7f800000u; // +inf
    memcpy(&lo, &floor_v, 4);
    unsigned long long* bad; uint32_t* first;
    (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 4);
    (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 4);
    hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, lo, hi, bad, first);
    unsigned long long h_bad = 0; uint32_t h_first = 0;
    (void)hipMemcpy(&h_bad, bad, 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(&h_first, first, 4, hipMemcpyDeviceToHost);
    printf("log10_pos vs log10f over %llu floats [0x%08x, 0x%08x]: %llu differ", (unsigned long long)hi - lo + 1, lo, hi, h_bad);
    if (h_bad) printf(" (first 0x%08x)", h_first);
    printf("\n");
    return h_bad ? 1 : 0;
}
