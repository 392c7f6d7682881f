// Correctly rounded float square root and reciprocal for arguments known to be NORMAL numbers of moderate size.
// The compiler's IEEE expansions (-fhip-fp32-correctly-rounded-divide-sqrt) also cover zeros, subnormals, infinities
// and NaNs, which costs them about twice the instructions; these sequences cover only [2^kFastExpLo, 2^(kFastExpHi+1))
// and are checked against the correctly rounded result for EVERY float of that range on the device itself
// (tools/fp/verify_fast_fp.hip, tests/test_gpu_fastfp.py).  Callers fall back to the IEEE expansion outside it.
#pragma once
#include <hip/hip_runtime.h>

namespace pt {

constexpr int kFastExpLo = -60, kFastExpHi = 60;

// true iff x is a positive normal float with kFastExpLo <= exponent <= kFastExpHi
__device__ __forceinline__ bool fast_fp_ok(float x) {
    const uint32_t e = __float_as_uint(x) >> 23;   // sign + exponent: a set sign bit makes it large
    return e - static_cast<uint32_t>(kFastExpLo + 127) <= static_cast<uint32_t>(kFastExpHi - kFastExpLo);
}

// RN(sqrt(x)): Goldschmidt step on the hardware reciprocal square root, then one exact-residual correction.
__device__ __forceinline__ float sqrt_rn_normal(float x) {
    const float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    const float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    const float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// RN(1 / x): hardware reciprocal, one Newton step, then the quotient corrected with its exact residual.
__device__ __forceinline__ float rcp_rn_normal(float x) {
    float r = __builtin_amdgcn_rcpf(x);
    const float e = __builtin_fmaf(-x, r, 1.0f);
    r = __builtin_fmaf(e, r, r);
    const float e2 = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e2, r, r);
}

}  // namespace pt
