// k_gemm_util.hpp — device helpers shared by the GEMM kernels (k_gemm.hip, k_gemm256.hip).
#pragma once
#include "vt_common.hpp"

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
    __builtin_amdgcn_global_load_lds(
        (const __attribute__((address_space(1))) void*)gsrc,
        (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// GELU(x) = x Phi(x) = max(x, 0) - |x| Phi(-|x|)   (exact identity for both signs).
// The Gaussian tail is evaluated as Phi(-a) = 2^P(a) with a degree-5 polynomial P fitted to
// log2(0.5 erfc(a / sqrt 2)) on [0, 9] (weighted so that the error of a * Phi(-a) is minimax): the
// absolute error of GELU is <= 6.4e-7 for every finite x (float32 evaluation, checked on 2e6
// points in [-30, 30]; beyond 9 the polynomial keeps falling, so the term underflows to 0 as the
// true tail does). Cost: 5 fma + v_exp + v_max + fma. The previous form (Abramowitz-Stegun erf:
// v_rcp + v_exp + 13 more VALU) took 8 us of a 256x256 fc1 tile round of 33 us, because a tile's
// epilogue is not overlapped with anything when a workgroup owns the whole CU.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fabsf(x);
    float p = __builtin_fmaf(-0.0004733149544335902f, a, 0.007084596436470747f);
    p = __builtin_fmaf(p, a, -0.05182747542858124f);
    p = __builtin_fmaf(p, a, -0.45999234914779663f);
    p = __builtin_fmaf(p, a, -1.1507878303527832f);
    p = __builtin_fmaf(p, a, -1.000037670135498f);
    return __builtin_fmaf(-a, __builtin_amdgcn_exp2f(p), fmaxf(x, 0.0f));
}

// The same arithmetic on two elements at once: the five polynomial steps and the final fma become
// v_pk_fma_f32 (two f32 per lane and issue slot - the packed form is what the CU's f32 peak rate is
// quoted on), each element's result is bit-identical to gelu_erf's. v_exp_f32, |x| and max stay
// per element (no packed forms).
__device__ __forceinline__ f32v2_t gelu_erf2(f32v2_t x) {
    const f32v2_t a = {fabsf(x.x), fabsf(x.y)};
    f32v2_t p = __builtin_elementwise_fma(f32v2_t{-0.0004733149544335902f, -0.0004733149544335902f}, a,
                                          f32v2_t{0.007084596436470747f, 0.007084596436470747f});
    p = __builtin_elementwise_fma(p, a, f32v2_t{-0.05182747542858124f, -0.05182747542858124f});
    p = __builtin_elementwise_fma(p, a, f32v2_t{-0.45999234914779663f, -0.45999234914779663f});
    p = __builtin_elementwise_fma(p, a, f32v2_t{-1.1507878303527832f, -1.1507878303527832f});
    p = __builtin_elementwise_fma(p, a, f32v2_t{-1.000037670135498f, -1.000037670135498f});
    const f32v2_t e = {__builtin_amdgcn_exp2f(p.x), __builtin_amdgcn_exp2f(p.y)};
    const f32v2_t r = {fmaxf(x.x, 0.0f), fmaxf(x.y, 0.0f)};
    return __builtin_elementwise_fma(-a, e, r);
}

// gelu_erf2 on FOUR element pairs in lock step (step k of every pair before step k + 1 of any): a packed fma
// result needs a wait state before the next packed fma reads it, and v_exp before its use; written pair after
// pair hipcc pads every dependent step with s_nop (216 per 128 elements in the persistent GEMM's epilogue,
// a section bound by vector issue). Each element's result is bit-identical to gelu_erf's.
__device__ __forceinline__ void gelu_erf2x4(f32v2_t (&x)[4]) {
    f32v2_t a[4], p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = f32v2_t{fabsf(x[i].x), fabsf(x[i].y)};
#pragma unroll
    for (int i = 0; i < 4; ++i)
        p[i] = __builtin_elementwise_fma(f32v2_t{-0.0004733149544335902f, -0.0004733149544335902f}, a[i],
                                         f32v2_t{0.007084596436470747f, 0.007084596436470747f});
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], a[i], f32v2_t{-0.05182747542858124f, -0.05182747542858124f});
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], a[i], f32v2_t{-0.45999234914779663f, -0.45999234914779663f});
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], a[i], f32v2_t{-1.1507878303527832f, -1.1507878303527832f});
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = __builtin_elementwise_fma(p[i], a[i], f32v2_t{-1.000037670135498f, -1.000037670135498f});
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = f32v2_t{__builtin_amdgcn_exp2f(p[i].x), __builtin_amdgcn_exp2f(p[i].y)};
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = __builtin_elementwise_fma(-a[i], p[i], f32v2_t{fmaxf(x[i].x, 0.0f), fmaxf(x[i].y, 0.0f)});
}

// A 4-B global load the compiler neither moves nor waits for: the caller puts an
// `s_waitcnt vmcnt(..)` (with the destination as an in/out operand) before the first use and does not
// touch the value in between.
__device__ __forceinline__ float gload_f32_asm(const float* p) {
    float v;
    asm volatile("global_load_dword %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// A 16-B global load with the same contract.
__device__ __forceinline__ u32x4_t gload_b128_asm(const void* p) {
    u32x4_t v;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}
// the same for 8 bytes (a lane's eight lo8 bytes of the residual pair)
__device__ __forceinline__ u32x2_t gload_b64_asm(const void* p) {
    u32x2_t v;
    asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// sum over the four lanes of a quad (lanes 4k .. 4k+3), every lane gets the same bits: two DPP
// quad_perm moves, no LDS traffic
// (full EXEC required: see half_wave_sum in vt_common.hpp)
__device__ __forceinline__ float quad_sum(float v) {
    const float a = v + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    return a + __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, a), 0x4E, 0xF, 0xF, true));
}

// ---- the X-epilogues' row-wise write-out: 8 consecutive columns of one row per lane ---------------
// x[0..7]: the float32 residual values. Emits the chunk partials of the 32-column chunk the lane's quad
// covers (quad-uniform: the four lanes of a quad hold the four 8-column groups of one chunk of one row),
// and the 3-byte pair. Fixed summation order -> run-to-run identical.
__device__ __forceinline__ void x_chunk_stats(const float (&x)[8], float& sum, float& m2) {
    float s = ((x[0] + x[1]) + (x[2] + x[3])) + ((x[4] + x[5]) + (x[6] + x[7]));
    s = quad_sum(s);
    const float mc = s * (1.0f / VT_STAT_CHUNK);
    float q = 0.0f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = x[e] - mc; q = __builtin_fmaf(d, d, q); }
    sum = s;
    m2 = quad_sum(q);
}
// x_split8 / x_join8: the 3-byte residual pair, vt_common.hpp (shared with the LayerNorm readers)

// value of the same lane of the other wave half (lane ^ 32): v_permlane32_swap, no LDS round trip
__device__ __forceinline__ float other_half_f32(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}

// Row terms (rstd, -mean * rstd) of a folded LayerNorm for the 4-wave kernel's bf16 epilogues, where the
// two lanes l and l + 32 hold the same row m: taken from the finalized array, or - small batches, where a
// launch of their own would cost more than the GEMM gains - combined in the epilogue from the producing
// X-epilogue's chunk partials, each of the two lanes taking half of the row's K / 32 chunks
// (parallel-variance form, as launch_rowstat_finalize). The partials (up to 16 chunks per lane: K <= 1024)
// are fetched BEFORE the main loop with inline-asm loads (hipcc would sink plain loads behind the loop and
// wait for them one by one: measured + 4 us on a 19 us GEMM); the caller waits vmcnt(0) after the loop -
// ln_wait - before ln_finish touches them. Both lanes of a row must take part.
struct LnPart { u32x4_t v[8]; };
__device__ __forceinline__ void ln_prefetch(const GemmArgs& p, int m, int half, LnPart& part) {
    if (p.rowstat) {
        part.v[0] = gload_b128_asm(reinterpret_cast<const char*>(p.rowstat + m) - ((m & 1) ? 8 : 0));   // 16-B aligned pair
        return;
    }
    const int nl = (p.K / VT_STAT_CHUNK) >> 2;          // 16-B loads per lane: two chunks each
    const char* src = reinterpret_cast<const char*>(p.cstat_in + (size_t)m * (p.K / VT_STAT_CHUNK)) + half * (nl * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < nl) part.v[i] = gload_b128_asm(src + i * 16);
}
// vm_drain: an operand-free wait in front of every wait that names registers. hipcc may place register
// copies of a "+v" operand directly in front of its asm statement (it does, for LnPart under control
// flow); with the drain ahead of them such a copy reads a retired load. tests/test_isa_asm_loads.py checks
// the emitted ISA for exactly this.
__device__ __forceinline__ void vm_drain() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);          // the scheduler would hoist those copies over an operand-free asm
}
__device__ __forceinline__ void ln_wait(LnPart& a) {
    vm_drain();
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(a.v[0]), "+v"(a.v[1]), "+v"(a.v[2]), "+v"(a.v[3]), "+v"(a.v[4]), "+v"(a.v[5]),
                 "+v"(a.v[6]), "+v"(a.v[7]) : : "memory");
}
__device__ __forceinline__ float2 ln_finish(const GemmArgs& p, int m, const LnPart& part) {
    if (p.rowstat)
        return (m & 1) ? make_float2(__uint_as_float(part.v[0][2]), __uint_as_float(part.v[0][3]))
                       : make_float2(__uint_as_float(part.v[0][0]), __uint_as_float(part.v[0][1]));
    const int nl = (p.K / VT_STAT_CHUNK) >> 2;
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < nl) s += __uint_as_float(part.v[i][0]) + __uint_as_float(part.v[i][2]);
    s += other_half_f32(s);
    const float Kf = (float)p.K, mean = s / Kf;
    float m2 = 0.0f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
        if (i < nl) {
            const float d0 = __uint_as_float(part.v[i][0]) * (1.0f / VT_STAT_CHUNK) - mean;
            const float d1 = __uint_as_float(part.v[i][2]) * (1.0f / VT_STAT_CHUNK) - mean;
            m2 += (__uint_as_float(part.v[i][1]) + (float)VT_STAT_CHUNK * (d0 * d0)) +
                  (__uint_as_float(part.v[i][3]) + (float)VT_STAT_CHUNK * (d1 * d1));
        }
    m2 += other_half_f32(m2);
    const float rstd = 1.0f / sqrtf(m2 / Kf + p.ln_eps);
    return make_float2(rstd, -mean * rstd);
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 63, "vmcnt immediate");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

