// k_attn.hip — joint template+search attention for gfx950 (head dim 64, N <= ~1k tokens).
//
// One wave owns 32 query rows of one (stream, head) and walks the keys in tiles of 32 with an
// online softmax. Orientation is chosen so that nothing crosses lanes except one half-swap per
// reduction:
//   S^T[key][q] = K_tile · Q^T        (MFMA A = K rows, B = Q rows)  -> the query is on the lane,
//                                      its 32 scores are in 16 registers x 2 lane halves
//   O^T[d][q]  += Vt_tile · P^T       (MFMA A = Vt rows, B = the S^T accumulator itself, converted
//                                      to bf16 in place: an accumulator tile is a valid B operand
//                                      for a product that sums over its ROW index)
// so running max / sum / rescale are per-lane scalars and P never goes through LDS.
// The k-order inside such a B operand is permuted (element j of lane half h is accumulator row
// 16s + 8(j>>2) + 4h + (j&3)); the Vt fragment is loaded in that same order, which with V stored
// transposed (Vt[b][h][d][key], written by the QKV GEMM epilogue) is two 8-B loads.
//
// Q and K are read straight from the row-major qk[M][2D] buffer (16 B per lane per k-step); Q is
// already scaled by 1/8. Keys >= tokens (padding of the last tile) are masked to -inf; Vt padding
// is zero.
#include "vt_common.hpp"

__device__ __forceinline__ bf16x8_t ld16(const bf16_t* p) {
    return *reinterpret_cast<const bf16x8_t*>(p);
}
__device__ __forceinline__ bf16x4_t ld8(const bf16_t* p) {
    return *reinterpret_cast<const bf16x4_t*>(p);
}
__device__ __forceinline__ float xhalf_max(float v) {
    return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float xhalf_sum(float v) {
    return v + __shfl_xor(v, 32);
}

__global__ __launch_bounds__(256) void attention_kernel(const bf16_t* __restrict__ qk,
                                                        const bf16_t* __restrict__ vt,
                                                        bf16_t* __restrict__ out, int tokens,
                                                        int H, int npad) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int qb = blockIdx.x * 4 + wave;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64, ld = 2 * D;
    const int nqb = (tokens + 31) >> 5;
    if (qb >= nqb) return;  // whole wave exits; no barriers in this kernel

    const int q = qb * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;
    const bf16_t* qrow = qk + ((size_t)b * tokens + qc) * ld + h * 64 + half * 8;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld16(qrow + ks * 16);

    const bf16_t* kbase = qk + (size_t)b * tokens * ld + D + h * 64 + half * 8;
    // Vt rows d = dc*32 + l31 of this (b, h); key offset 4*half inside each 8-key group
    const bf16_t* vbase = vt + ((size_t)(b * H + h) * 64 + l31) * npad + 4 * half;

    f32x16_t o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.0f; o1[r] = 0.0f; }
    float m_run = -1.0e30f, l_run = 0.0f;

    const int nkt = (tokens + 31) >> 5;
    for (int kt = 0; kt < nkt; ++kt) {
        const int key = kt * 32 + l31;
        const int kc = key < tokens ? key : tokens - 1;
        const bf16_t* krow = kbase + (size_t)kc * ld;
        bf16x8_t kf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) kf[ks] = ld16(krow + ks * 16);
        // V fragments for this key tile: [dc][s2] = 8 keys in the permuted order
        bf16x8_t vf[2][2];
#pragma unroll
        for (int dc = 0; dc < 2; ++dc)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16_t* vp = vbase + (size_t)dc * 32 * npad + kt * 32 + 16 * s2;
                bf16x4_t lo = ld8(vp), hi = ld8(vp + 8);
                vf[dc][s2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }

        f32x16_t s;
#pragma unroll
        for (int r = 0; r < 16; ++r) s[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
            s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);

        // s[r]: key = kt*32 + (r&3) + 8*(r>>2) + 4*half, query = this lane's
        float mx = -1.0e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kidx = kt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (kidx >= tokens) s[r] = -INFINITY;
            mx = fmaxf(mx, s[r]);
        }
        mx = xhalf_max(mx);
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __expf(m_run - m_new);
        float psum = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            p[r] = __expf(s[r] - m_new);
            psum += p[r];
        }
        psum = xhalf_sum(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }

        bf16x8_t pf[2];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            union { uint32_t u[4]; bf16x8_t v; } cv;
#pragma unroll
            for (int e = 0; e < 4; ++e)
                cv.u[e] = pack_bf16x2(p[8 * s2 + 2 * e], p[8 * s2 + 2 * e + 1]);
            pf[s2] = cv.v;
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][s2], pf[s2], o0, 0, 0, 0);
            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][s2], pf[s2], o1, 0, 0, 0);
        }
    }

    if (q < tokens) {
        const float inv = 1.0f / l_run;
        bf16_t* orow = out + ((size_t)b * tokens + q) * D + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // registers 4g..4g+3 hold d = 8g + 4*half + (0..3)
            const int d = 8 * g + 4 * half;
            uint2 a = make_uint2(pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                 pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            uint2 c = make_uint2(pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                 pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            *reinterpret_cast<uint2*>(orow + d) = a;
            *reinterpret_cast<uint2*>(orow + 32 + d) = c;
        }
    }
}

hipError_t launch_attention(const bf16_t* qk, const bf16_t* vt, bf16_t* out, int B, int tokens,
                            int H, int npad, hipStream_t st) {
    const int nqb = (tokens + 31) / 32;
    dim3 grid((nqb + 3) / 4, H, B);
    hipLaunchKernelGGL(attention_kernel, grid, dim3(256), 0, st, qk, vt, out, tokens, H, npad);
    return hipGetLastError();
}
