// nn_device.h -- device-side helpers shared by the NSNet2 kernels (kernels_nn.hip, kernels_ws.hip)
#pragma once
#include <hip/hip_runtime.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// GRU geometry: H = 400 hidden units = 25 unit tiles of 16 = 25 super-steps of 16 (NSNet2-baseline)
constexpr int GRU_H = 400;
constexpr int GRU_J = GRU_H / 16;
constexpr int GRU2_SLAB = 3 * GRU_J * 256; // floats of one unit tile's recurrent weights [3 g][25 S][64][4]: 75 KB

// gate nonlinearities on v_exp_f32 / v_rcp_f32
__device__ __forceinline__ float fast_sigmoid(float x)
{
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * -1.44269504088896341f));
}
__device__ __forceinline__ float fast_tanh(float x)
{
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(x * 2.88539008177792681f));
}
