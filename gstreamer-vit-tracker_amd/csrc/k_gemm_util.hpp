// k_gemm_util.hpp — device helpers shared by the GEMM kernels (k_gemm.hip, k_gemm256.hip).
#pragma once
#include "vt_common.hpp"

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// GELU(x) = 0.5 x (1 + erf(x / sqrt 2)). erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7,
// far below the bf16 rounding of the result): one v_rcp, one v_exp and a degree-5 polynomial
// instead of libm's erff (~3x the instructions), which made the fc1 epilogue a visible share of
// the kernel at large tiles.
__device__ __forceinline__ float gelu_erf(float x) {
    const float z = fabsf(x) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, z, 1.0f));
    float poly = __builtin_fmaf(1.061405429f, t, -1.453152027f);
    poly = __builtin_fmaf(poly, t, 1.421413741f);
    poly = __builtin_fmaf(poly, t, -0.284496736f);
    poly = __builtin_fmaf(poly, t, 0.254829592f);
    // erf(|x|/sqrt2) = 1 - poly*t*exp(-z^2); exp(-z^2) = 2^(-z^2 log2 e)
    const float g = poly * t * __builtin_amdgcn_exp2f(-1.4426950408889634f * z * z);
    const float erfv = __builtin_copysignf(1.0f - g, x);
    const float hx = 0.5f * x;
    return __builtin_fmaf(hx, erfv, hx);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

