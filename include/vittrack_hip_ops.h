/*
 * vittrack_hip_ops.h — operator-level entry points of libvittrack_hip_ops.so: numerics tests and tuning
 * tools call the same kernel launchers (same objects) the tracker pass uses, on operands of their choosing.
 * NOT part of the product boundary: libvittrack_hip.so (include/vittrack_hip.h, the drop-in for the
 * reference's vit_tracker crate, src/tracker_context.rs:21,88,90,120) exports none of these; the ops
 * library is that library's objects plus csrc/vt_ops.hip and exports the boundary as well, so one handle
 * serves a test that needs both.
 */
#ifndef VITTRACK_HIP_OPS_H
#define VITTRACK_HIP_OPS_H

#include "vittrack_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- operator-level entry points (numerics tests call the same kernels the pass uses) ---- */

/* acc[M,N] = A[M,K] (bf16 bits) x W[N,K]^T (bf16 bits), float32 accumulation. Host pointers.
 * K % 64 == 0, N % 64 == 0. cfg: tile configuration as in vt_op_gemm_bench (< 0: the launcher's own
 * choice for the shape). epilogue:
 *   0  x = acc + bias                    the X-epilogues the engine keeps its residual stream with: x is
 *   1  x = (acc + bias) + c_inout        stored as the 3-byte pair (hi = bf16(x), lo8 = clamp(rint((x - hi) * 2^12), +-127))
 *   4  x = (acc + bias) + pos            and comes back as hi + lo8 * 2^-12; pos = c_inout, one row per
 *                                        output row. rowstat_out (may be NULL) receives per row the terms
 *                                        (rstd, -mean * rstd) of LayerNorm(x) with `eps`, computed from the
 *                                        float32 x before the split (what the consuming GEMM multiplies with).
 *   2  GELU(y) -> bf16, 3  ReLU(y) -> bf16 (returned widened to f32); y = acc + bias, or with a folded
 *      LayerNorm (rowstat_in [M][2] and colsum [N] not NULL): y = rowstat_in[m][0] * acc +
 *      (rowstat_in[m][1] * colsum[n] + bias[n]). */
int vt_op_gemm_bf16(int device_id, const uint16_t* a, const uint16_t* w, const float* bias,
                    float* c_inout, int M, int N, int K, int epilogue, int cfg,
                    const float* rowstat_in, const float* colsum, float* rowstat_out, float eps);
/* Kernel-tuning helper: mean microseconds per launch of the GEMM kernel on device-resident random
 * operands. epilogue uses the library's internal numbering (0 f32+pos, 1 residual, 2 GELU, 3 ReLU,
 * 4 QKV, 5 f32); cfg: 0 = 64x64 ring 4, 1 = 128x128 ring 3, 2 = 64x64 ring 2, 3 = 128x128 ring 2
 * (K-tile depth 64); 4 = 64x64 ring 3, 5 = 64x64 ring 2, 6 = 128x128 ring 2 (K-tile depth 128, K % 128 == 0);
 * 7 = 64x64 ring 3 and 8 = 128x64 ring 3 with four loader waves (K-tile depth 128);
 * 18 / 19 = 256x256 8-wave kernels (one tile per workgroup / persistent); <0 = the launcher's own choice.
 * Bits 8-9 of a cfg >= 0 (here, in vt_op_gemm_bf16 and in vt_op_qkv_bf16) force the order in which an XCD's run of
 * workgroups covers the tile grid of configurations 0-8: 0 = the launcher's rule, 1 = row panels x all columns,
 * 2 = column tiles x all rows (placement only: the results are the same bits). */
int vt_op_gemm_bench(int device_id, int M, int N, int K, int epilogue, int cfg, int iters,
                     float* us_out);
/* The head's 3x3 convolution (zero padding) + bias + ReLU as the engine runs it - an implicit GEMM whose
 * A loads gather the im2col row: t [B*grid*grid][C] bf16, w [N][9*C] bf16 (column (ky*3+kx)*C + c),
 * out [B*grid*grid][N] (bf16 widened to f32). C % 64 == 0, N % 64 == 0; cfg 0..8 (4..6: C % 128 == 0; 7, 8: the
 * loader-wave forms of 0 / 4), < 0: launcher's choice. */
int vt_op_conv3x3_relu_bf16(int device_id, const uint16_t* t, const uint16_t* w, const float* bias,
                            float* out, int B, int grid, int C, int N, int cfg);
/* The head's band kernel (csrc/k_head.hip) on its own: out = relu(conv(t) + bias). conv3x3 != 0: t [B*grid*grid][Cin],
 * w [N][9*Cin] (column (ky*3+kx)*Cin + c), N == Cin, zero padding; else the 1x1 layer, w [N][Cin]. R (rows of the
 * map per workgroup) / ncb (16-column blocks per wave) <= 0: the launcher's plan. t == NULL: operands filled with a
 * fixed pseudo-random pattern (timing runs). out (nullable): bf16 values widened to f32. iters > 0 and us_out: mean
 * microseconds per launch. */
int vt_op_headconv_bf16(int device_id, const uint16_t* t, const uint16_t* w, const float* bias, float* out,
                        int B, int grid, int Cin, int N, int conv3x3, int R, int ncb, int iters, float* us_out);
/* The head's first (1x1) layer with the final LayerNorm in front of it: out[b*grid*grid + cell][n] =
 * relu(LayerNorm(xh + xl * 2^-12)[b*ntok + off + cell] . w[n] + bias[n]); xh (bf16 bits) / xl (signed bytes) [B*ntok][D] the
 * 3-byte pair of the residual stream, gamma / beta [D], w [N][D], D = 768 or 1024. fused != 0: ONE launch - the band kernel normalises its band's
 * rows itself (what the engine runs); fused == 0: the LayerNorm kernel, then the band kernel on its output - the fused
 * form reproduces it bit for bit. R / ncb / iters / us_out / out as above; xh == NULL: synthetic operands (timing). */
int vt_op_headconv_ln_bf16(int device_id, const uint16_t* xh, const int8_t* xl, const float* gamma, const float* beta,
                           float eps, int ntok, int off, const uint16_t* w, const float* bias, float* out, int B, int grid,
                           int D, int N, int fused, int R, int ncb, int iters, float* us_out);
/* The QKV projection with its attention-layout epilogue: a [B*tokens, D], w [3D, D], bias [3D] ->
 * qk_out [B*tokens, 2D] (q scaled by 1/8, then k) and vt_out [B*H, 64, npad] (v transposed per head,
 * npad = tokens rounded up to 64, padding zero); bf16 results widened to f32. cfg as above;
 * vt_perm = 1: Vt in the key order attention mode 3 reads (attn_perm16 inside every 16 keys).
 * rowstat_in [B*tokens][2] / colsum [3D] (both or neither NULL): a folded LayerNorm as in vt_op_gemm_bf16. */
int vt_op_qkv_bf16(int device_id, const uint16_t* a, const uint16_t* w, const float* bias,
                   float* qk_out, float* vt_out, int B, int tokens, int D, int cfg, int vt_perm,
                   const float* rowstat_in, const float* colsum);
/* out[B,N,H*64] (bf16 widened to f32) = softmax(q k^T) v per head; q,k,v: [B,N,H*64] bf16 bits
 * (q already scaled). mode as in vt_op_attention_bench. */
int vt_op_attention_bf16(int device_id, const uint16_t* q, const uint16_t* k, const uint16_t* v,
                         float* out, int B, int N, int H, int mode);
/* Kernel-tuning helper: mean microseconds per launch of the attention kernel on random data;
 * mode 0 key-split, 1 independent waves, 2 LDS-shared tiles, 3 LDS-DMA ring (permuted Vt),
 * <0 the launcher's choice. */
int vt_op_attention_bench(int device_id, int B, int N, int H, int mode, int iters, float* us_out);
/* Kernel-timing helper: mean microseconds per launch of the whole-frame NV12 -> RGB8 converter
 * (the reference's nv12_full_to_rgb_parallel, src/nv12_convert.rs:46-92) on a device-resident
 * w x h frame of random bytes (HIP events around `iters` launches). Algorithmic traffic is
 * 1.5 + 3 bytes per pixel. */
int vt_op_nv12_to_rgb8_bench(int device_id, int w, int h, int iters, float* us_out);
/* The same for vt_nv12_to_rgb8_batch_device: n frames of w x h converted by one launch per 64 frames. */
int vt_op_nv12_to_rgb8_batch_bench(int device_id, int w, int h, int n, int iters, float* us_out);
/* y[M,D] (bf16 widened) = LayerNorm(x[M,D] f32; gamma, beta, eps=1e-6) */
int vt_op_layernorm(int device_id, const float* x, const float* gamma, const float* beta,
                    float* y, int M, int D);

#ifdef __cplusplus
}
#endif
#endif /* VITTRACK_HIP_OPS_H */
