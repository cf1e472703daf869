// k_misc.hip — LayerNorm, row statistics and the LayerNorm fold for gfx950 (the head lives in k_head.hip).
#include "vt_common.hpp"

// ---- LayerNorm: one wave per row, row kept in registers, two-pass variance -----------------------
// Residual stream in (float32, or SPLIT: the 3-byte pair of the engine, x = xh + xl * 2^-12: vt_common.hpp), bf16 GEMM
// operand out. In the engine only the FINAL LayerNorm (search tokens, before the head) runs as a kernel:
// the two LayerNorms of every block are folded into the GEMMs that consume them (vt_common.hpp).
// NCH = D / 128 float2 chunks per lane.
template <int NCH, bool SPLIT>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x,
                                                        const bf16_t* __restrict__ xh,
                                                        const uint8_t* __restrict__ xl,
                                                        const float* __restrict__ gamma,
                                                        const float* __restrict__ beta,
                                                        bf16_t* __restrict__ y, int rows, int D,
                                                        int group, int in_stride, int in_off,
                                                        float eps) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const size_t in_row = (size_t)(r / group) * in_stride + in_off + (r % group);
    float2 v[NCH];
    float sum = 0.0f;
    if constexpr (SPLIT) {
        const uint32_t* hr = reinterpret_cast<const uint32_t*>(xh + in_row * D);
        const uint16_t* lr = reinterpret_cast<const uint16_t*>(xl + in_row * D);      // two lo8 bytes per lane
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const uint32_t h = hr[lane + 64 * j], l = lr[lane + 64 * j];
            v[j].x = __builtin_fmaf(lo8_f32(l, 0), VT_LO_Q, __uint_as_float(h << 16));
            v[j].y = __builtin_fmaf(lo8_f32(l, 1), VT_LO_Q, __uint_as_float(h & 0xffff0000u));
        }
    } else {
        const float2* xr = reinterpret_cast<const float2*>(x + in_row * D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) v[j] = xr[lane + 64 * j];
    }
    const float2* g2 = reinterpret_cast<const float2*>(gamma);     // with the row, not after the reductions
    const float2* b2 = reinterpret_cast<const float2*>(beta);
    float2 gq[NCH], bq[NCH];
#pragma unroll
    for (int j = 0; j < NCH; ++j) { gq[j] = g2[lane + 64 * j]; bq[j] = b2[lane + 64 * j]; }
#pragma unroll
    for (int j = 0; j < NCH; ++j)      // keep the loads up here (hipcc sinks them behind the reductions)
        asm volatile("" : "+v"(gq[j].x), "+v"(gq[j].y), "+v"(bq[j].x), "+v"(bq[j].y) : : "memory");
#pragma unroll
    for (int j = 0; j < NCH; ++j) sum += v[j].x + v[j].y;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sum += __shfl_xor(sum, o);
    const float mean = sum / (float)D;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        v[j].x -= mean; v[j].y -= mean;
        sq += v[j].x * v[j].x + v[j].y * v[j].y;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    const float rstd = 1.0f / sqrtf(sq / (float)D + eps);
    uint32_t* yr = reinterpret_cast<uint32_t*>(y + (size_t)r * D);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const float2 g = gq[j], b = bq[j];
        yr[lane + 64 * j] = pack_bf16x2((v[j].x * rstd) * g.x + b.x, (v[j].y * rstd) * g.y + b.y);
    }
}

// Wide variant for D % 256 == 0 (the ViT-B / ViT-L widths): half a wave per row, each lane owns
// NCH chunks of 8 consecutive floats -> 16-B loads and 16-B bf16 stores (the one-wave-per-row kernel
// above stores 4 B per lane, 256 B per wave instruction). Same two-pass arithmetic per row; the
// summation order inside a row differs from the narrow kernel (both are fixed, run-to-run stable).
template <int NCH, bool SPLIT>
__global__ __launch_bounds__(256) void layernorm_wide_kernel(const float* __restrict__ x,
                                                             const bf16_t* __restrict__ xh,
                                                             const uint8_t* __restrict__ xl,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             bf16_t* __restrict__ y, int rows, int D,
                                                             int group, int in_stride, int in_off,
                                                             float eps) {
    const int l32 = threadIdx.x & 31;
    int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = r < rows;
    r = live ? r : rows - 1;                  // idle half-waves redo the last row, store nothing
    const size_t in_row = (size_t)(r / group) * in_stride + in_off + (r % group);
    f32x4_t v[NCH][2];
    if constexpr (SPLIT) {
        const u32x4_t* hr = reinterpret_cast<const u32x4_t*>(xh + in_row * D);
        const u32x2_t* lr = reinterpret_cast<const u32x2_t*>(xl + in_row * D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) ln_unpack_split(hr[l32 + 32 * j], lr[l32 + 32 * j], v[j]);
    } else {
        const f32x4_t* xr = reinterpret_cast<const f32x4_t*>(x + in_row * D);
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            v[j][0] = xr[2 * (l32 + 32 * j)];
            v[j][1] = xr[2 * (l32 + 32 * j) + 1];
        }
    }
    // gamma / beta do not depend on the row: fetched together with it, not after the two reductions
    // (one dependent memory round trip less - what a launch of a few hundred rows is made of)
    LnCoef<NCH> k;
    ln_load_coef<NCH>(gamma, beta, l32, k);
    uint4 o[NCH];
    ln_row<NCH>(v, k, D, eps, o);
    uint4* yr = reinterpret_cast<uint4*>(y + (size_t)r * D);
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        if (live) yr[l32 + 32 * j] = o[j];
}

template <bool SPLIT>
static hipError_t launch_layernorm_any(const float* x, const bf16_t* xh, const uint8_t* xl, const float* gamma,
                                       const float* beta, bf16_t* y, int rows, int D, int group, int in_stride,
                                       int in_off, float eps, hipStream_t st) {
    if (rows <= 0 || D % 128 != 0) return hipErrorInvalidValue;
    if (D % 256 == 0 && D / 256 <= 4) {
        dim3 gridw((rows + 7) / 8), block(256);
#define LNW_CASE(n)                                                                                      \
    case n:                                                                                              \
        vt_launch((layernorm_wide_kernel<n, SPLIT>), gridw, block, 0, st, x, xh, xl, gamma, beta, y, \
                           rows, D, group, in_stride, in_off, eps);                                      \
        break;
        switch (D / 256) { LNW_CASE(1) LNW_CASE(2) LNW_CASE(3) LNW_CASE(4) }
#undef LNW_CASE
        return hipGetLastError();
    }
    dim3 grid((rows + 3) / 4), block(256);
#define LN_CASE(n)                                                                                       \
    case n:                                                                                              \
        vt_launch((layernorm_kernel<n, SPLIT>), grid, block, 0, st, x, xh, xl, gamma, beta, y, rows, \
                           D, group, in_stride, in_off, eps);                                            \
        break;
    switch (D / 128) {
        LN_CASE(1) LN_CASE(2) LN_CASE(3) LN_CASE(4) LN_CASE(6) LN_CASE(8) LN_CASE(10) LN_CASE(12)
        default: return hipErrorInvalidValue;
    }
#undef LN_CASE
    return hipGetLastError();
}

hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, bf16_t* y,
                            int rows, int D, int group, int in_stride, int in_off, float eps,
                            hipStream_t st) {
    return launch_layernorm_any<false>(x, nullptr, nullptr, gamma, beta, y, rows, D, group, in_stride, in_off, eps, st);
}

hipError_t launch_layernorm_split(const bf16_t* xh, const uint8_t* xl, const float* gamma, const float* beta,
                                  bf16_t* y, int rows, int D, int group, int in_stride, int in_off,
                                  float eps, hipStream_t st) {
    return launch_layernorm_any<true>(nullptr, xh, xl, gamma, beta, y, rows, D, group, in_stride, in_off, eps, st);
}

// ---- row statistics of the residual stream from the X-epilogues' chunk partials --------------------
// cstat[m][c] = (sum, M2 about the chunk mean) of the 32 columns of chunk c (float32 value before the
// bf16 split). Half a wave per row combines them (parallel-variance form, no cancellation):
//     mean = sum_c s_c / D,   M2 = sum_c (q_c + 32 (s_c / 32 - mean)^2),   rstd = 1 / sqrt(M2 / D + eps)
// and stores what the consuming GEMM's epilogue multiplies with: (rstd, -mean * rstd).
__global__ __launch_bounds__(256) void rowstat_finalize_kernel(const float2* __restrict__ cstat,
                                                               float2* __restrict__ rowstat, int M,
                                                               int nchunk, float eps) {
    const int l32 = threadIdx.x & 31;
    int r = blockIdx.x * 8 + (threadIdx.x >> 5);
    const bool live = r < M;
    r = live ? r : M - 1;
    const float2* row = cstat + (size_t)r * nchunk;
    float2 c0 = make_float2(0.0f, 0.0f), c1 = c0;
    const bool h0 = l32 < nchunk, h1 = l32 + 32 < nchunk;      // nchunk <= 64 (D <= 2048)
    if (h0) c0 = row[l32];
    if (h1) c1 = row[l32 + 32];
    float s = c0.x + c1.x;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    const float D = (float)(nchunk * VT_STAT_CHUNK);
    const float mean = s / D;
    const float d0 = c0.x * (1.0f / VT_STAT_CHUNK) - mean, d1 = c1.x * (1.0f / VT_STAT_CHUNK) - mean;
    float m2 = (h0 ? c0.y + (float)VT_STAT_CHUNK * (d0 * d0) : 0.0f) + (h1 ? c1.y + (float)VT_STAT_CHUNK * (d1 * d1) : 0.0f);
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) m2 += __shfl_xor(m2, o);
    const float rstd = 1.0f / sqrtf(m2 / D + eps);
    if (live && l32 == 0) rowstat[r] = make_float2(rstd, -mean * rstd);
}

hipError_t launch_rowstat_finalize(const float2* cstat, float2* rowstat, int M, int nchunk, float eps,
                                   hipStream_t st) {
    if (M <= 0 || nchunk < 1 || nchunk > 64) return hipErrorInvalidValue;
    vt_launch(rowstat_finalize_kernel, dim3((M + 7) / 8), dim3(256), 0, st, cstat, rowstat, M, nchunk, eps);
    return hipGetLastError();
}

// ---- LayerNorm folded into the consuming GEMM's weights (once per engine) ---------------------------
// One wave per output feature n: Wf[n][k] = bf16(gamma[k] * W[n][k]), colsum[n] = sum_k Wf[n][k],
// cvec[n] = sum_k beta[k] W[n][k] + bias[n]. Fixed lane-strided summation order.
__global__ __launch_bounds__(256) void fold_layernorm_kernel(const bf16_t* __restrict__ W,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta,
                                                             const float* __restrict__ bias,
                                                             bf16_t* __restrict__ Wf, float* __restrict__ colsum,
                                                             float* __restrict__ cvec, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const uint32_t* wr = reinterpret_cast<const uint32_t*>(W + (size_t)n * K);
    uint32_t* wo = reinterpret_cast<uint32_t*>(Wf + (size_t)n * K);
    float s = 0.0f, c = 0.0f;
    for (int k2 = lane; k2 < K / 2; k2 += 64) {
        const uint32_t pk = wr[k2];
        const float w0 = __uint_as_float(pk << 16), w1 = __uint_as_float(pk & 0xffff0000u);
        const uint32_t f = pack_bf16x2(gamma[2 * k2] * w0, gamma[2 * k2 + 1] * w1);
        wo[k2] = f;
        s += __uint_as_float(f << 16) + __uint_as_float(f & 0xffff0000u);
        c += beta[2 * k2] * w0 + beta[2 * k2 + 1] * w1;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) { s += __shfl_xor(s, o); c += __shfl_xor(c, o); }
    if (lane == 0) { colsum[n] = s; cvec[n] = c + bias[n]; }
}

hipError_t launch_fold_layernorm(const bf16_t* W, const float* gamma, const float* beta, const float* bias,
                                 bf16_t* Wf, float* colsum, float* cvec, int N, int K, hipStream_t st) {
    if (N <= 0 || K <= 0 || (K & 1)) return hipErrorInvalidValue;
    vt_launch(fold_layernorm_kernel, dim3((N + 3) / 4), dim3(256), 0, st, W, gamma, beta, bias, Wf, colsum,
                       cvec, N, K);
    return hipGetLastError();
}
