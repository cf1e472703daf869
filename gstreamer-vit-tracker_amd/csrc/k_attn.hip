// k_attn.hip — joint template+search attention for gfx950 (head dim 64, N <= ~1k tokens).
//
// One wave owns 32 query rows of one (stream, head) and walks key tiles of 32 with an online
// softmax. Orientation is chosen so that nothing crosses lanes except one half-swap per reduction:
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
// already scaled by 1/8. The fragments of the next key tile are loaded before the current tile is
// computed (register double buffer), so L2 latency overlaps the MFMAs and the softmax.
//
// Two work splits (same arithmetic per tile):
//   KVSPLIT = false: the 4 waves of a block take 4 different query blocks (many streams: enough
//                    blocks to fill the chip, no merge).
//   KVSPLIT = true : the 4 waves of a block share ONE query block and take every 4th key tile;
//                    the partial (max, sum, O) are merged through LDS. With one stream there are
//                    only 12 x 23 query blocks for 256 CUs; splitting the keys gives 4x the waves
//                    and a 4x shorter dependent chain per wave.
// Keys >= tokens (padding of the last tile) are masked to -inf; Vt padding is zero.
#include <cstdlib>

#include "vt_common.hpp"
#include "k_gemm_util.hpp"

__device__ __forceinline__ bf16x8_t ld16(const bf16_t* p) {
    return *reinterpret_cast<const bf16x8_t*>(p);
}
__device__ __forceinline__ bf16x4_t ld8(const bf16_t* p) {
    return *reinterpret_cast<const bf16x4_t*>(p);
}
// combine a value with the same lane of the other wave half: v_permlane32_swap (no LDS round trip)
__device__ __forceinline__ float other_half(float v) {
    const unsigned u = __float_as_uint(v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    // r[0]: lanes 32-63 now hold the lower half's value; r[1]: lanes 0-31 hold the upper half's
    return __uint_as_float((threadIdx.x & 32) ? r[0] : r[1]);
}
__device__ __forceinline__ float xhalf_max(float v) { return fmaxf(v, other_half(v)); }
__device__ __forceinline__ float xhalf_sum(float v) { return v + other_half(v); }


// One 32-key sub-tile for one wave: s = S^T accumulator (key on the register axis, query on the
// lane). Updates the running max / sum, rescales O only when the max moved, and multiplies P into
// O^T. VALU budget matters here (the tile is 8 MFMAs): exp2 with the log2(e) factor folded into
// one fma, hardware bf16 packing, and no O rescale while the running max is unchanged.
// Lazy running maximum: m_run is only moved when a score exceeds it by more than ATT_LAZY_LOG2 (in
// log2 units), so probabilities are at most 2^8 (harmless in bf16 / f32) and the rescale branch -
// 32 multiplies, and in modes 3/4 also a pass over the scores - is not taken every time one of the
// wave's 32 queries sees a slightly larger score (with random-like scores that is most steps).
#define ATT_LAZY_LOG2 8.0f

// Scores arrive in log2 units (the QKV epilogue scales q by log2(e)/8), so p = 2^(s - m).
// The row sum is taken over the bf16-ROUNDED probabilities, the values the P·V product actually
// uses (numerator and denominator then round alike; mode 3 gets this sum from an MFMA).
__device__ __forceinline__ void softmax_pv(f32x16_t s, const bf16x8_t (&vf)[2][2], f32x16_t& o0,
                                           f32x16_t& o1, float& m_run, float& l_run) {
    float mx = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[2], s[3]));
#pragma unroll
    for (int r = 4; r < 16; r += 2) mx = fmaxf(mx, fmaxf(s[r], s[r + 1]));
    mx = xhalf_max(mx);
    if (!__all(mx <= m_run + ATT_LAZY_LOG2)) {   // some query's max grew by more than 2^8: rescale (wave-uniform branch)
        const float m_new = fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        m_run = m_new;
    }
    float psum = 0.0f;
    bf16x8_t pf[2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        union { uint32_t u[4]; bf16x8_t v; } cv;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const uint32_t u = pack_bf16x2(__builtin_amdgcn_exp2f(s[8 * s2 + 2 * e] - m_run),
                                           __builtin_amdgcn_exp2f(s[8 * s2 + 2 * e + 1] - m_run));
            psum += __uint_as_float(u << 16) + __uint_as_float(u & 0xffff0000u);
            cv.u[e] = u;
        }
        pf[s2] = cv.v;
    }
    l_run += xhalf_sum(psum);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[0][s2], pf[s2], o0, 0, 0, 0);
        o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[1][s2], pf[s2], o1, 0, 0, 0);
    }
}

// S^T tile = K_tile · Q^T (4 k-steps over d = 64). MASK: -inf on keys >= tokens (only the last
// tile of a sequence is instantiated with MASK, so full tiles carry no compare/select work).
template <bool MASK>
__device__ __forceinline__ f32x16_t qk_scores(const bf16x8_t (&kf)[4], const bf16x8_t (&qf)[4],
                                              int key0, int tokens, int half) {
    const f32x16_t zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                           0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    f32x16_t s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0], qf[0], zero, 0, 0, 0);
#pragma unroll
    for (int ks = 1; ks < 4; ++ks)
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf[ks], s, 0, 0, 0);
    if (MASK) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (key0 + (r & 3) + 8 * (r >> 2) + 4 * half >= tokens) s[r] = -INFINITY;
    }
    return s;
}

struct KvFrag {
    bf16x8_t k[4];      // K rows of the tile, 4 k-steps over d
    bf16x8_t v[2][2];   // Vt rows [d chunk][key half], keys in the accumulator-operand order
};

__device__ __forceinline__ void load_tile(KvFrag& f, const bf16_t* kbase, const bf16_t* vbase,
                                          int kt, int tokens, int ld, int npad, int l31) {
    const int key = kt * 32 + l31;
    const int kc = key < tokens ? key : tokens - 1;
    const bf16_t* krow = kbase + (size_t)kc * ld;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) f.k[ks] = ld16(krow + ks * 16);
#pragma unroll
    for (int dc = 0; dc < 2; ++dc)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const bf16_t* vp = vbase + (size_t)dc * 32 * npad + kt * 32 + 16 * s2;
            const bf16x4_t lo = ld8(vp), hi = ld8(vp + 8);
            f.v[dc][s2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        }
}

template <bool KVSPLIT>
__global__ __launch_bounds__(256, 2) void attention_kernel(const bf16_t* __restrict__ qk,
                                                        const bf16_t* __restrict__ vt,
                                                        bf16_t* __restrict__ out, int tokens,
                                                        int H, int npad) {
    __shared__ float s_o[KVSPLIT ? 4 * 32 * 64 : 1];
    __shared__ float s_ml[KVSPLIT ? 4 * 2 * 64 : 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64, ld = 2 * D;
    const int nqb = (tokens + 31) >> 5;
    const int qb = KVSPLIT ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
    if (!KVSPLIT && qb >= nqb) return;  // whole wave exits; no barrier on this path

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
    const int kt0 = KVSPLIT ? wave : 0, kstep = KVSPLIT ? 4 : 1;
    KvFrag cur, nxt;
    if (kt0 < nkt) load_tile(cur, kbase, vbase, kt0, tokens, ld, npad, l31);
    for (int kt = kt0; kt < nkt; kt += kstep) {
        // prefetch the next tile of this wave (clamped: the last iteration reloads a valid tile)
        const int ktn = (kt + kstep < nkt) ? kt + kstep : kt;
        load_tile(nxt, kbase, vbase, ktn, tokens, ld, npad, l31);

        // s[r]: key = kt*32 + (r&3) + 8*(r>>2) + 4*half, query = this lane's
        const f32x16_t s = (kt * 32 + 32 > tokens)
                               ? qk_scores<true>(cur.k, qf, kt * 32, tokens, half)
                               : qk_scores<false>(cur.k, qf, kt * 32, tokens, half);
        softmax_pv(s, cur.v, o0, o1, m_run, l_run);
        cur = nxt;
    }

    bf16_t* orow = out + ((size_t)b * tokens + q) * D + h * 64;
    if constexpr (!KVSPLIT) {
        if (q < tokens) {
            const float inv = 1.0f / l_run;
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
    } else {
        // merge the 4 key-split partials: O = sum_w O_w e^{m_w - m} / sum_w l_w e^{m_w - m}
        float* so = s_o + wave * (32 * 64);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            so[r * 64 + lane] = o0[r];
            so[(16 + r) * 64 + lane] = o1[r];
        }
        s_ml[(wave * 2 + 0) * 64 + lane] = m_run;
        s_ml[(wave * 2 + 1) * 64 + lane] = l_run;
        __syncthreads();
        float mw[4], m = -1.0e30f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            mw[w] = s_ml[(w * 2 + 0) * 64 + lane];
            m = fmaxf(m, mw[w]);
        }
        float l = 0.0f, sc[4];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            sc[w] = __builtin_amdgcn_exp2f(mw[w] - m);
            l += s_ml[(w * 2 + 1) * 64 + lane] * sc[w];
        }
        const float inv = 1.0f / l;
        // wave w finalises accumulator registers [8w, 8w+8) of the 32 (two 4-register groups)
#pragma unroll
        for (int gg = 0; gg < 2; ++gg) {
            const int r0 = 8 * wave + 4 * gg;          // 0..28, step 4
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float acc = 0.0f;
#pragma unroll
                for (int w = 0; w < 4; ++w) acc += s_o[w * (32 * 64) + (r0 + e) * 64 + lane] * sc[w];
                v[e] = acc * inv;
            }
            // register index r0 -> chunk (r0 >= 16), group g = (r0 & 15) / 4 -> d = 8g + 4*half
            const int d = (r0 >= 16 ? 32 : 0) + 8 * ((r0 & 15) >> 2) + 4 * half;
            if (q < tokens)
                *reinterpret_cast<uint2*>(orow + d) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
        }
    }
}

// ---- many streams: 4 query blocks per workgroup share the K / Vt tiles through LDS ---------------------
// 64-key tiles, register-staged (loads of tile t+1 are issued before tile t is computed and written
// to the other LDS buffer afterwards: one barrier per tile). K image: 128-B rows with the GEMM's
// chunk swizzle (c ^ ((row >> 1) & 7)) -> conflict-free ds_read_b128; Vt image: 64 rows (d) of 64
// keys padded to 136 B -> the two 8-B reads per fragment hit 64 distinct banks.
#define ATT_KT 64
#define ATT_K_BYTES (ATT_KT * 128)
#define ATT_V_STRIDE 136
#define ATT_V_BYTES (64 * ATT_V_STRIDE)

__global__ __launch_bounds__(256, 2) void attention_lds_kernel(const bf16_t* __restrict__ qk,
                                                            const bf16_t* __restrict__ vt,
                                                            bf16_t* __restrict__ out, int tokens,
                                                            int H, int npad) {
    __shared__ __attribute__((aligned(16))) char smem[2 * (ATT_K_BYTES + ATT_V_BYTES)];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int h = blockIdx.y, b = blockIdx.z;
    const int D = H * 64, ld = 2 * D;
    const int nqb = (tokens + 31) >> 5;
    const int qb = blockIdx.x * 4 + wave;
    const bool active = qb < nqb;          // inactive waves still stage tiles and hit barriers

    const int q = qb * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;
    const bf16_t* qrow = qk + ((size_t)b * tokens + qc) * ld + h * 64 + half * 8;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld16(qrow + ks * 16);

    // staging: each thread moves two 16-B chunks of the K tile and two of the Vt tile
    const bf16_t* kg = qk + (size_t)b * tokens * ld + D + h * 64;
    const bf16_t* vg = vt + (size_t)(b * H + h) * 64 * npad;
    // chunk c = tid + 256 j (j = 0, 1): row c >> 3, 16-B piece c & 7. Named scalars, not arrays
    // captured by a lambda: the latter kept the staging registers in scratch memory.
    const int c0 = tid, c1 = tid + 256;
    const int krow0 = c0 >> 3, krow1 = c1 >> 3, kch0 = c0 & 7, kch1 = c1 & 7;
    const int koff0 = krow0 * 128 + ((kch0 ^ ((krow0 >> 1) & 7)) << 4);
    const int koff1 = krow1 * 128 + ((kch1 ^ ((krow1 >> 1) & 7)) << 4);
    const int voff0 = krow0 * ATT_V_STRIDE + kch0 * 16, voff1 = krow1 * ATT_V_STRIDE + kch1 * 16;
    const bf16_t* vsrc0 = vg + (size_t)krow0 * npad + kch0 * 8;
    const bf16_t* vsrc1 = vg + (size_t)krow1 * npad + kch1 * 8;
    uint4 kreg0, kreg1, vreg0, vreg1;
#define ATT_GLOAD(KT)                                                                        \
    {                                                                                        \
        int key_a = (KT) * ATT_KT + krow0, key_b = (KT) * ATT_KT + krow1;                    \
        key_a = key_a < tokens ? key_a : tokens - 1;                                         \
        key_b = key_b < tokens ? key_b : tokens - 1;                                         \
        kreg0 = *reinterpret_cast<const uint4*>(kg + (size_t)key_a * ld + kch0 * 8);         \
        kreg1 = *reinterpret_cast<const uint4*>(kg + (size_t)key_b * ld + kch1 * 8);         \
        vreg0 = *reinterpret_cast<const uint4*>(vsrc0 + (KT) * ATT_KT);                      \
        vreg1 = *reinterpret_cast<const uint4*>(vsrc1 + (KT) * ATT_KT);                      \
    }
#define ATT_LSTORE(BUF)                                                                      \
    {                                                                                        \
        char* sk_ = smem + (BUF) * (ATT_K_BYTES + ATT_V_BYTES);                              \
        char* sv_ = sk_ + ATT_K_BYTES;                                                       \
        *reinterpret_cast<uint4*>(sk_ + koff0) = kreg0;                                      \
        *reinterpret_cast<uint4*>(sk_ + koff1) = kreg1;                                      \
        *reinterpret_cast<uint2*>(sv_ + voff0) = make_uint2(vreg0.x, vreg0.y);               \
        *reinterpret_cast<uint2*>(sv_ + voff0 + 8) = make_uint2(vreg0.z, vreg0.w);           \
        *reinterpret_cast<uint2*>(sv_ + voff1) = make_uint2(vreg1.x, vreg1.y);               \
        *reinterpret_cast<uint2*>(sv_ + voff1 + 8) = make_uint2(vreg1.z, vreg1.w);           \
    }

    f32x16_t o0, o1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.0f; o1[r] = 0.0f; }
    float m_run = -1.0e30f, l_run = 0.0f;

    const int nt = (tokens + ATT_KT - 1) / ATT_KT;
    ATT_GLOAD(0)
    ATT_LSTORE(0)
    __syncthreads();
    for (int kt = 0; kt < nt; ++kt) {
        if (kt + 1 < nt) ATT_GLOAD(kt + 1)
        const char* sk = smem + (kt & 1) * (ATT_K_BYTES + ATT_V_BYTES);
        const char* sv = sk + ATT_K_BYTES;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int key0 = kt * ATT_KT + st * 32;
            if (key0 >= tokens) break;     // block-uniform
            bf16x8_t kf[4], vf[2][2];
            const int row = st * 32 + l31;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
                kf[ks] = *reinterpret_cast<const bf16x8_t*>(
                    sk + row * 128 + (((2 * ks + half) ^ ((row >> 1) & 7)) << 4));
#pragma unroll
            for (int dc = 0; dc < 2; ++dc)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const char* vp = sv + (dc * 32 + l31) * ATT_V_STRIDE +
                                     (st * 32 + 16 * s2 + 4 * half) * 2;
                    const bf16x4_t lo = *reinterpret_cast<const bf16x4_t*>(vp);
                    const bf16x4_t hi = *reinterpret_cast<const bf16x4_t*>(vp + 16);
                    vf[dc][s2] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            const f32x16_t s = (key0 + 32 > tokens)
                                   ? qk_scores<true>(kf, qf, key0, tokens, half)
                                   : qk_scores<false>(kf, qf, key0, tokens, half);
            softmax_pv(s, vf, o0, o1, m_run, l_run);
        }
        if (kt + 1 < nt) ATT_LSTORE((kt + 1) & 1)
        __syncthreads();
    }

    if (active && q < tokens) {
        const float inv = 1.0f / l_run;
        bf16_t* orow = out + ((size_t)b * tokens + q) * D + h * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
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

// ---- mode 3: LDS-DMA ring, 64-key steps, permuted Vt ------------------------------------------------
// Same arithmetic as above (32 queries per wave, S^T = K Q^T, O^T += Vt P^T with the S accumulator
// as the B operand), restructured around what the PMC counters of mode 2 showed: the kernel is
// VALU-bound (58 % VALU busy, 22 % MFMA busy, 18.5 VALU instructions per MFMA, about twice what
// the softmax itself needs). Changes:
//   * K and Vt tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4) into a 3-stage ring with
//     a counted vmcnt and ONE raw barrier per 64-key step: no staging registers, no ds_write, no
//     per-step clamping arithmetic.
//   * Vt is stored by the QKV epilogue with the two middle 4-key runs of every 16 keys swapped
//     (attn_perm16), so the Vt fragment of a lane - keys {4h..4h+3, 8+4h..8+4h+3} of a 16-key
//     group, the k-order of an accumulator used as B operand - is ONE ds_read_b128 at chunk
//     2*s2 + h: K and Vt tiles then have the same LDS image ([64 rows][128 B], chunk c of row r
//     at c ^ ((r >> 1) & 7)) and share their four per-lane read addresses.
//   * One softmax per 64 keys: v_max3 reductions, one cross-half exchange and one lazy-rescale
//     test per step instead of per 32 keys.
//   * The row sums come from a third P·V MFMA against a tile of ones (the matrix pipe was 28 % busy):
//     they are the sums of the bf16-rounded probabilities, the values the numerator uses.
//   * No running-maximum bookkeeping in the common case: scores are used as they come out of the
//     MFMA (p = 2^s) while they stay within +-32 log2 units; the reference moves (with a rescale)
//     only when a query's maximum leaves that window. With 32 queries per wave a plain "max grew"
//     test took the rescale branch in almost every step, and the per-step subtraction of the
//     running maximum was 32 of the ~115 VALU instructions of a step.
//   Measured and dropped (B = 30, 720 tokens, 12 heads; this kernel 74 us = 650 TFLOP/s, mode 2
//   103 us): feeding -m_run as the C operand of the first QK MFMA to save the VALU subtraction
//   (+16 VGPRs -> 2 instead of 3 waves per SIMD: 81 us); an 8-wave variant with the two wave groups
//   one barrier apart, MFMA phase against softmax phase (110 us: in-kernel stamps gave 1320 cycles
//   for the 20-MFMA phase beside the partner's 790-cycle softmax phase, plus 2 x 260 cycles in the
//   barriers); an XCD-contiguous block order alone changed nothing (K/Vt re-reads are L2 hits);
//   software-pipelining the step inside the wave (QK of step kt+1 and P·V of step kt-1 issued
//   among the softmax VALU of step kt, two register sets, 250 VGPRs -> 2 waves per SIMD: 84 us).
//   In-kernel stamps of this kernel, cycles per 64-key step and wave (3,530 in all): vmcnt + barrier
//   276; LDS-DMA issue + 16 ds_read + QK (behind the previous step's P·V still in the pipe) + max
//   1,920; 2^x + pack 607; P·V issue 418 - a wave's step is latency, not issue, and three waves per
//   SIMD hide about two thirds of it.
//   * The output tile is transposed through LDS and stored as whole 128-B rows.
#define AT3_STAGE 16384
#define ATT_WIN 32.0f          // half-width of the score window inside which the reference stays put
#define AT3_NS 3

// acc + the sum of the 8 bf16 values of a P fragment: four v_dot2c_f32_bf16 against (1, 1) - exact products,
// float32 accumulation, i.e. the sum of the ROUNDED probabilities the P.V product uses
__device__ __forceinline__ float sum_p8(const bf16x8_t& p, float acc) {
    typedef __bf16 bf2_t __attribute__((ext_vector_type(2)));
    union { bf16x8_t v; uint32_t u[4]; } c;
    c.v = p;
    const bf2_t one2 = __builtin_bit_cast(bf2_t, 0x3f803f80u);
#pragma unroll
    for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf2_t, c.u[e]), one2, acc, false);
    return acc;
}

__device__ __forceinline__ float max3f(float a, float b, float c) {
    return __builtin_fmaxf(__builtin_fmaxf(a, b), c);   // selected as v_max3_f32
}

// AT3_NS = 3 ring stages of 16 KiB: two tiles ahead, 48 KiB -> 3 workgroups per CU.
// One pass of a workgroup over its (stream, head, 128 queries). CAREFUL = false: the FIRST
// pass takes p = 2^score as it is, with no running maximum at all - per step 16 v_max3, a cross-half
// swap, a ballot and a branch less, and the exponentials start the moment the scores leave the matrix
// pipe. Afterwards every query's row sum must lie in [2^-60, 2^60] (it does whenever the scores stayed
// within about +-60 log2 units; an overflow gives inf or NaN, which fail the test too); if any query of
// the workgroup fails, nothing is stored and the function returns true: the kernel then runs the
// CAREFUL pass (the windowed reference described below) on the same block. While no score leaves the
// +-32 window the two passes are the same arithmetic. Two inlined copies, one after the other, rather
// than a loop with a flag: with a back edge hipcc keeps the whole set-up live (190 VGPRs, 98 spilled
// SGPRs: two waves per SIMD instead of three).
// The row sums are kept on the vector ALU (sum_p8: 16 v_dot2c per 64-key step and lane, the
// two lane halves added once at the end) instead of a third P.V MFMA against a tile of ones - 4 of the 20
// MFMAs of a step; without the running-maximum bookkeeping the step is bound by the matrix pipe, not by
// vector issue (profiles/r03_attention_ab.txt).
template <bool CAREFUL>
__device__ __forceinline__ bool at3_pass(char* smem, const bf16_t* __restrict__ qk, const bf16_t* __restrict__ vt,
                                         bf16_t* __restrict__ out, int tokens, int H, int npad, int tid, int block) {
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int D = H * 64, ld = 2 * D;
    const int nqb = (tokens + 31) >> 5, nxb = (nqb + 3) >> 2;
    // Workgroups are dealt round-robin over the 8 XCDs. The nxb query blocks of one (stream, head)
    // read the same K and Vt (184 KB at 720 tokens): give each XCD a contiguous run of the 1-D grid
    // so that they share one L2 instead of pulling the head's K/Vt into six of them (measured
    // before this remap: the kernel ran at the HBM rate of 6x the K/V bytes).
    int bid = block;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int xb = bid % nxb, bh = bid / nxb;
    const int h = bh % H, b = bh / H;
    const int qb = xb * 4 + wave;

    f32x16_t o0, o1;
    constexpr bool careful = CAREFUL;
    const int q = qb * 32 + l31;
    const int qc = q < tokens ? q : tokens - 1;     // idle rows repeat the last query, never stored
    const bf16_t* qrow = qk + ((size_t)b * tokens + qc) * ld + h * 64 + half * 8;
    bf16x8_t qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) qf[ks] = ld16(qrow + ks * 16);

    // staging: LDS position (g, tid) of a tile = row g*32 + (tid >> 3), stored chunk tid & 7, which
    // holds logical chunk (tid & 7) ^ ((row >> 1) & 7)
    const int nt = (tokens + 63) >> 6;
    const int srow = tid >> 3, schunk = (tid & 7) ^ ((tid >> 4) & 7);
    const char* kg = reinterpret_cast<const char*>(qk + (size_t)b * tokens * ld + D + h * 64 + schunk * 8);
    const char* vg = reinterpret_cast<const char*>(vt + (size_t)(b * H + h) * 64 * npad + schunk * 8);
    const uint32_t koff0 = (uint32_t)srow * ld * 2, koff1 = (uint32_t)(srow + 32) * ld * 2;
    const uint32_t voff0 = (uint32_t)srow * npad * 2, voff1 = (uint32_t)(srow + 32) * npad * 2;
    // the last tile may reach past this stream's keys: clamp its rows (masked below)
    const int lk0 = (nt - 1) * 64 + srow, lk1 = lk0 + 32;
    const uint32_t klast0 = (uint32_t)(lk0 < tokens ? lk0 : tokens - 1) * ld * 2;
    const uint32_t klast1 = (uint32_t)(lk1 < tokens ? lk1 : tokens - 1) * ld * 2;
#define AT3_STAGE_TILE(KT, SBASE)                                                          \
    {                                                                                      \
        const bool last_ = (KT) == nt - 1;                                                 \
        const char* kp_ = kg + (last_ ? 0 : (size_t)(KT) * 64 * ld * 2);                   \
        const char* vp_ = vg + (size_t)(KT) * 128;                                         \
        char* d_ = smem + (SBASE) + wave * 1024;                                           \
        glds16(kp_ + (last_ ? klast0 : koff0), d_);                                        \
        glds16(kp_ + (last_ ? klast1 : koff1), d_ + 4096);                                 \
        glds16(vp_ + voff0, d_ + 8192);                                                    \
        glds16(vp_ + voff1, d_ + 12288);                                                   \
    }

    // fragment read addresses inside a stage: row l31 (+32 per tile half / d chunk), chunk
    // (2*ks + half) ^ sw; the same four serve K (offset st*4096) and Vt (8192 + dc*4096)
    const int sw = (l31 >> 1) & 7;
    const uint32_t fa0 = (uint32_t)(l31 * 128 + (((0 + half) ^ sw) << 4));
    const uint32_t fa1 = (uint32_t)(l31 * 128 + (((2 + half) ^ sw) << 4));
    const uint32_t fa2 = (uint32_t)(l31 * 128 + (((4 + half) ^ sw) << 4));
    const uint32_t fa3 = (uint32_t)(l31 * 128 + (((6 + half) ^ sw) << 4));

    // O^T accumulators.
    // Softmax reference: p = 2^(score - m_run). m_run stays 0 - no subtraction pass at all - while
    // every query's scores stay inside [-ATT_WIN, +ATT_WIN] log2 units (p <= 2^32 is harmless in
    // bf16 / f32 and the normalisation at the end divides it out); it moves, with a rescale of what
    // has been accumulated, only when a maximum leaves that window (upwards in any step; downwards
    // in the first step, so that a row of uniformly tiny scores does not underflow).
    float m_run = 0.0f;
    float lsum = 0.0f;                   // this lane's share of its query's row sum (the two lane halves are added at the end)
    bool shifted = false;                // wave-uniform: some lane's m_run != 0
    constexpr int NS = AT3_NS;
    constexpr int AHEAD = NS - 1;        // tiles in flight beyond the one being computed
#pragma unroll
    for (int r = 0; r < 16; ++r) { o0[r] = 0.0f; o1[r] = 0.0f; }
    AT3_STAGE_TILE(0, 0)
    if (AHEAD > 1 && nt > 1) AT3_STAGE_TILE(1, AT3_STAGE)
    int sbase = 0;                       // LDS offset of the stage holding tile kt
    // The unchecked pass does not compute what is thrown away: a wave whose 32 queries lie past the last
    // token (the fourth wave of a stream's last workgroup at 720 tokens) only stages its share of the tiles,
    // and when at most 32 keys of the last tile exist (720 = 11 x 64 + 16) that tile is a half step after
    // the loop: one QK chain, 16 exponentials, 6 of the 12 P.V MFMAs (4 % of the kernel's work each).
    const bool active = CAREFUL || qb < nqb;
    const bool half_last = !CAREFUL && (tokens & 63) != 0 && (tokens & 63) <= 32;
    const int nfull = half_last ? nt - 1 : nt;
    for (int kt = 0; kt < nfull; ++kt) {
        if (AHEAD > 1 && kt + 1 < nt) wait_vmcnt<4>(); else wait_vmcnt<0>();
        // every wave's pieces of tile kt have landed, and every wave is done with tile kt-1,
        // whose stage tile kt+AHEAD overwrites
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (kt + AHEAD < nt) {
            int s2base = sbase + AHEAD * AT3_STAGE;
            s2base = s2base >= NS * AT3_STAGE ? s2base - NS * AT3_STAGE : s2base;
            AT3_STAGE_TILE(kt + AHEAD, s2base)
        }
        if (!active) {                    // wave-uniform
            sbase = sbase + AT3_STAGE >= NS * AT3_STAGE ? 0 : sbase + AT3_STAGE;
            continue;
        }
        const char* st = smem + sbase;
        const char* p0 = st + fa0; const char* p1 = st + fa1;
        const char* p2 = st + fa2; const char* p3 = st + fa3;
#define AT3_RD(P, OFF) (*reinterpret_cast<const bf16x8_t*>((P) + (OFF)))
        bf16x8_t kf0[4], kf1[4], vf0[4], vf1[4];
        kf0[0] = AT3_RD(p0, 0); kf0[1] = AT3_RD(p1, 0); kf0[2] = AT3_RD(p2, 0); kf0[3] = AT3_RD(p3, 0);
        kf1[0] = AT3_RD(p0, 4096); kf1[1] = AT3_RD(p1, 4096); kf1[2] = AT3_RD(p2, 4096); kf1[3] = AT3_RD(p3, 4096);
        vf0[0] = AT3_RD(p0, 8192); vf0[1] = AT3_RD(p1, 8192); vf0[2] = AT3_RD(p2, 8192); vf0[3] = AT3_RD(p3, 8192);
        vf1[0] = AT3_RD(p0, 12288); vf1[1] = AT3_RD(p1, 12288); vf1[2] = AT3_RD(p2, 12288); vf1[3] = AT3_RD(p3, 12288);

        // s = K Q^T - m_run (log2 units): two independent accumulation chains
        const f32x16_t zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                               0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        f32x16_t s0, s1;
        s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[0], qf[0], zero, 0, 0, 0);
        s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[0], qf[0], zero, 0, 0, 0);
#pragma unroll
        for (int ks = 1; ks < 4; ++ks) {
            s0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf0[ks], qf[ks], s0, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf1[ks], qf[ks], s1, 0, 0, 0);
        }
        if (kt == nt - 1 && (tokens & 63) != 0) {      // block-uniform: keys >= tokens -> -inf
                const int key0 = kt * 64 + 4 * half;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int key = key0 + (r & 3) + 8 * (r >> 2);
                    if (key >= tokens) s0[r] = -INFINITY;
                    if (key + 32 >= tokens) s1[r] = -INFINITY;
                }
            }
            if constexpr (careful) {
            if (shifted) {                      // wave-uniform, rare: scores relative to the moved reference
    #pragma unroll
                for (int r = 0; r < 16; ++r) { s0[r] -= m_run; s1[r] -= m_run; }
            }
            // ---- online softmax over the 64 keys of the step (per lane: one query, 32 scores) ----
            float mx = max3f(s0[0], s0[1], s0[2]);
    #pragma unroll
            for (int r = 3; r < 15; r += 2) mx = max3f(mx, s0[r], s0[r + 1]);
            mx = max3f(mx, s0[15], s1[0]);
    #pragma unroll
            for (int r = 1; r < 15; r += 2) mx = max3f(mx, s1[r], s1[r + 1]);
            mx = fmaxf(mx, s1[15]);
            mx = xhalf_max(mx);                 // max of (score - m_run) over the query's 64 keys
            const bool first = kt == 0;
            if (!__all(mx <= ATT_WIN) || (first && !__all(mx >= -ATT_WIN))) {   // wave-uniform, rare
                const float dm = first ? mx : fmaxf(mx, 0.0f);          // m_new - m_run, per lane
                const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-dm);   // O and the sum are 0 in step 0
                m_run += dm;
                shifted = true;
                lsum *= alpha;
    #pragma unroll
                for (int r = 0; r < 16; ++r) {
                    o0[r] *= alpha; o1[r] *= alpha;
                    s0[r] -= dm; s1[r] -= dm;
                }
            }
            }   // careful
            bf16x8_t pf[4];
    #pragma unroll
            for (int g = 0; g < 2; ++g) {
                union { uint32_t u[4]; bf16x8_t v; } c0, c1;
    #pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c0.u[e] = pack_bf16x2(__builtin_amdgcn_exp2f(s0[8 * g + 2 * e]), __builtin_amdgcn_exp2f(s0[8 * g + 2 * e + 1]));
                    c1.u[e] = pack_bf16x2(__builtin_amdgcn_exp2f(s1[8 * g + 2 * e]), __builtin_amdgcn_exp2f(s1[8 * g + 2 * e + 1]));
                }
                pf[g] = c0.v;
                pf[2 + g] = c1.v;
            }
    #pragma unroll
            for (int g = 0; g < 4; ++g) {       // 16-key groups of the step
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf0[g], pf[g], o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf1[g], pf[g], o1, 0, 0, 0);
                lsum = sum_p8(pf[g], lsum);
            }
        sbase = sbase + AT3_STAGE >= NS * AT3_STAGE ? 0 : sbase + AT3_STAGE;
    }
    if constexpr (!CAREFUL) {
        if (half_last) {                  // block-uniform: keys nt*64-64 .. tokens-1 (1 .. 32 of them)
            wait_vmcnt<0>();
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (active) {
                const char* st = smem + sbase;
                const char* p0 = st + fa0; const char* p1 = st + fa1;
                const char* p2 = st + fa2; const char* p3 = st + fa3;
                const f32x16_t zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f,
                                       0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                f32x16_t sh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p0, 0), qf[0], zero, 0, 0, 0);
                sh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p1, 0), qf[1], sh, 0, 0, 0);
                sh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p2, 0), qf[2], sh, 0, 0, 0);
                sh = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p3, 0), qf[3], sh, 0, 0, 0);
                const int key0 = (nt - 1) * 64 + 4 * half;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (key0 + (r & 3) + 8 * (r >> 2) >= tokens) sh[r] = -INFINITY;
                union { uint32_t u[4]; bf16x8_t v; } c0, c1;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    c0.u[e] = pack_bf16x2(__builtin_amdgcn_exp2f(sh[2 * e]), __builtin_amdgcn_exp2f(sh[2 * e + 1]));
                    c1.u[e] = pack_bf16x2(__builtin_amdgcn_exp2f(sh[8 + 2 * e]), __builtin_amdgcn_exp2f(sh[8 + 2 * e + 1]));
                }
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p0, 8192), c0.v, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p0, 12288), c0.v, o1, 0, 0, 0);
                lsum = sum_p8(c0.v, lsum);
                o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p1, 8192), c1.v, o0, 0, 0, 0);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AT3_RD(p1, 12288), c1.v, o1, 0, 0, 0);
                lsum = sum_p8(c1.v, lsum);
            }
        }
    }
    __syncthreads();                      // every wave is done reading the ring
    if constexpr (!careful) {
        // did every query of the workgroup stay in range? (the barriers of a pass are workgroup-wide, so
        // the four waves repeat together or not at all; the flags live in the dead ring)
        const float l = xhalf_sum(lsum);
        const bool bad = active && !(l >= 0x1p-60f && l <= 0x1p60f);
        const unsigned long long bm = __ballot(bad);
        int* flag = reinterpret_cast<int*>(smem);
        if (lane == 0) flag[wave] = bm != 0ull;
        __syncthreads();
        const int any = flag[0] | flag[1] | flag[2] | flag[3];
        __syncthreads();                  // flags read before anything overwrites them
        if (any) return true;
    }
    const float l_run = xhalf_sum(lsum);

    // ---- epilogue: O^T (d on registers, query on lanes) -> LDS [32 q][128 B] per wave -> rows ----
    {
        const float inv = 1.0f / l_run;
        char* ow = smem + wave * 4096;     // 32 rows x 128 B, chunk c of row r at c ^ (r & 7)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // registers 4g..4g+3 hold d = 8g + 4*half + (0..3): 8 B at byte 16g + 8*half (+64 for o1)
            const uint2 a = make_uint2(pack_bf16x2(o0[4 * g] * inv, o0[4 * g + 1] * inv),
                                       pack_bf16x2(o0[4 * g + 2] * inv, o0[4 * g + 3] * inv));
            const uint2 c = make_uint2(pack_bf16x2(o1[4 * g] * inv, o1[4 * g + 1] * inv),
                                       pack_bf16x2(o1[4 * g + 2] * inv, o1[4 * g + 3] * inv));
            *reinterpret_cast<uint2*>(ow + l31 * 128 + ((g ^ (l31 & 7)) << 4) + 8 * half) = a;
            *reinterpret_cast<uint2*>(ow + l31 * 128 + (((4 + g) ^ (l31 & 7)) << 4) + 8 * half) = c;
        }
        // same wave reads back what it wrote (no block barrier needed); 8 lanes per row
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int r = it * 8 + (lane >> 3), c = lane & 7;
            const uint4 v = *reinterpret_cast<const uint4*>(ow + r * 128 + ((c ^ (r & 7)) << 4));
            const int qq = qb * 32 + r;
            if (qb < nqb && qq < tokens)
                *reinterpret_cast<uint4*>(out + ((size_t)b * tokens + qq) * D + h * 64 + c * 8) = v;
        }
    }
    return false;
}

__global__ __launch_bounds__(256, 3) void attention_dma_kernel(const bf16_t* __restrict__ qk,
                                                            const bf16_t* __restrict__ vt,
                                                            bf16_t* __restrict__ out, int tokens,
                                                            int H, int npad) {
    __shared__ __attribute__((aligned(16))) char smem[AT3_NS * AT3_STAGE];
    // launch bound of 3 waves per SIMD: the first pass fits 168 registers without a spill; what hipcc then spills
    // (100 B of scratch) sits in the rare second pass only (checked in the ISA)
    if (at3_pass<false>(smem, qk, vt, out, tokens, H, npad, (int)threadIdx.x, (int)blockIdx.x)) {
        __syncthreads();
        // the (rare) second pass rebuilds every address from opaque copies of its inputs: if the
        // compiler could see that the two inlined passes compute the same values it would keep the
        // first pass's alive for the second (189 VGPRs instead of 156: a wave per SIMD less)
        int t2 = threadIdx.x, b2 = blockIdx.x, tk = tokens, hh = H, np2 = npad;
        const bf16_t* q2 = qk; const bf16_t* v2 = vt; bf16_t* o2 = out;
        asm volatile("" : "+v"(t2));
        asm volatile("" : "+s"(b2), "+s"(tk), "+s"(hh), "+s"(np2), "+s"(q2), "+s"(v2), "+s"(o2));
        at3_pass<true>(smem, q2, v2, o2, tk, hh, np2, t2, b2);
    }
}

// mode: 0 = key-split (few streams), 1 = independent waves, 2 = LDS-shared tiles, register staged (the fallbacks for
// token counts the DMA kernel does not take), 3 = LDS-DMA ring with permuted Vt (default when tokens % 4 == 0 and
// npad % 64 == 0), -1 = choose. The alternatives of mode 3 that rounds 2-5 measured and dropped (one softmax per 32
// keys; sequential halves in <= 128 registers = four workgroups per CU; row sums by a ones-MFMA or by 4x4x4 MFMAs;
// the careful pass alone) are in git history with their numbers in profiles/r03_attention_ab.txt and
// profiles/r05_attention_modes.txt.
int attention_pick_mode(int tokens, int npad) {
    if (npad % 64 != 0) return 0;
    return (tokens % 4 == 0) ? 3 : 2;   // tokens % 4: the QKV epilogue's 4-token runs stay inside a stream
}

// once per device, outside any stream capture (nothing to raise at present: all LDS is static)
hipError_t attention_prepare() { return hipSuccess; }

hipError_t launch_attention_mode(const bf16_t* qk, const bf16_t* vt, bf16_t* out, int B, int tokens,
                                 int H, int npad, int mode, hipStream_t st) {
    const int nqb = (tokens + 31) / 32;
    if (mode < 0) mode = attention_pick_mode(tokens, npad);
    if (mode >= 2 && npad % 64 != 0) return hipErrorInvalidValue;
    if (mode >= 3 && tokens % 4 != 0) return hipErrorInvalidValue;
    if (mode == 0) {
        vt_launch(attention_kernel<true>, dim3(nqb, H, B), dim3(256), 0, st, qk, vt, out,
                           tokens, H, npad);
    } else if (mode == 1) {
        vt_launch(attention_kernel<false>, dim3((nqb + 3) / 4, H, B), dim3(256), 0, st, qk,
                           vt, out, tokens, H, npad);
    } else if (mode == 2) {
        vt_launch(attention_lds_kernel, dim3((nqb + 3) / 4, H, B), dim3(256), 0, st, qk, vt,
                           out, tokens, H, npad);
    } else if (mode == 3) {
        vt_launch(attention_dma_kernel, dim3(((nqb + 3) / 4) * H * B), dim3(256), 0, st, qk, vt, out, tokens, H, npad);
    } else {
        return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_attention(const bf16_t* qk, const bf16_t* vt, bf16_t* out, int B, int tokens,
                            int H, int npad, hipStream_t st) {
    return launch_attention_mode(qk, vt, out, B, tokens, H, npad, attention_pick_mode(tokens, npad), st);
}
