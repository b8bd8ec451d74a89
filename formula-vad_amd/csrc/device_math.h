// Device-side scalar helpers shared by the spectral kernels and the tools that check them.
#pragma once
#include <hip/hip_runtime.h>

// log10f for x >= 1e-12 (or +inf): what the device library's log10f computes for a normal argument -- v_log_f32, the product
// with log10(2) as a two-piece constant, inf passed through -- without its scaling of denormal arguments (7 instructions
// instead of 17; the same bits: tools/log10_check.hip compares the two over every float from 1e-12 up)
__device__ __forceinline__ float log10_pos(float x)
{
    const float y = __builtin_amdgcn_logf(x);
    const float c_hi = __int_as_float(0x3e9a209a), c_lo = __int_as_float(0x3284fbcf);
    const float r = y * c_hi;
    float t = __builtin_fmaf(y, c_hi, -r);
    t = __builtin_fmaf(c_lo, y, t);
    const float o = r + t;
    return __builtin_fabsf(y) < __builtin_inff() ? o : y;
}
