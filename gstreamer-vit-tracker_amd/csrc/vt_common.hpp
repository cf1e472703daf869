// vt_common.hpp — types shared by the host engine and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/vittrack_hip.h"

typedef uint16_t bf16_t;  // bfloat16 bit pattern
typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

// one device-resident input frame (mirrors vt_frame, 64-bit pointers)
struct FrameDesc {
    const uint8_t* p0;  // RGB8 packed pixels or NV12 Y plane
    const uint8_t* p1;  // NV12 interleaved UV plane
    int32_t w, h, s0, s1, fmt;
    int32_t x0, y0;     // frame coordinates of the first stored pixel (window upload)
    int32_t ww, wh;     // extent of the stored window in pixels: samples outside it read as black
    int32_t pad;
};
static_assert(sizeof(FrameDesc) == 56, "FrameDesc layout");

// per tracked stream, lives in HBM; the decode kernel of frame t writes what the preprocessing
// kernel of frame t+1 reads, so a stream of updates never needs the host in between.
struct StreamState {
    float box[4];        // last accepted box, frame pixels: x, y, w, h (top-left + size)
    float geo[4];        // search-crop geometry of the running frame: x0m, y0m, scale, side
    int32_t frame_w, frame_h;
    int32_t initialized;
    int32_t frames_done;     // updates completed since init
    int32_t success_count;   // of which result.success
    int32_t last_idx;        // argmax cell of the last update
    float last_fbox[4];      // unrounded clipped box of the last update
    float last_score;
    // index (= frames_done + 1 at the time) of the last pass in which the crop needed a pixel that
    // lies inside the frame but outside the window the caller stored: such samples read as black, so
    // a host that uploaded a SPECULATIVE window (vt_group_enqueue_host) must redo that pass
    int32_t window_miss;
    int32_t pad[2];
};

struct ModelDims {
    int patch, T, S, D, H, L, mlp, C, kpad;
    int gt, gs, nt, ns, ntok;   // grids and token counts
    int npad;                   // ntok rounded up to 64 (attention key padding)
    float norm_a[3], norm_b[3];
    float success_threshold, ln_eps;
};

// ---- kernel launchers (each enqueues on `st`; no allocation, no synchronisation) -------------

// q is stored multiplied by 1/sqrt(64) * log2(e), so that q·k is the softmax exponent in log2 units
// (the attention kernels use v_exp_f32 = 2^x directly); float(log2 e) / 8 is exact in float32
#define ATT_Q_SCALE (0.125f * 1.4426950408889634f)

// NUMERICAL SPECIFICATION v3 (round 6). The residual stream x is stored as a 3-BYTE pair: Xh = bf16(x) [M][ldx] and
// Xl = lo8 [M][ldx] bytes, lo8 = clamp(rint((x - Xh) * 2^12), -127, 127) as a signed integer: x = Xh + lo8 * 2^-12.
// Until round 5 the low half was a second bf16 (4 bytes per element, 17 significant bits); the byte plane moves a
// quarter less through the X-epilogues' read-modify-write and the final LayerNorm, the phases of the pass that are
// bound by memory requests (profiles/r06_lo8_residual.txt: whole frame + 2.0 %). x - Xh is exact (Xh is x rounded to 8
// significant bits), the scaling is a power of two, rint is round-to-nearest-even, Xh + lo8 * 2^-12 is exact in
// float32: oracle (oracle/vit_ref.py split_residual) and kernels can differ only through the x they start from.
// |x - Xh| <= ulp(Xh) / 2: at most 64 quanta for |x| < 8 (this model: |x| < 3.5) - the byte holds it with room to spare
// and the stored value is within half a quantum of x. For 8 <= |x| < 16 the remainder reaches 128 quanta next to a bf16
// tie, where the clamp costs at most one quantum; beyond 16 it leaves part of the low half behind - the value degrades
// towards plain bf16, it never wraps.
#define VT_LO_SHIFT 12
#define VT_LO_Q (1.0f / 4096.0f)
#define VT_LO_MAX (127.0f / 4096.0f)
// (comment of the v2 form, for the history of the format:)
// The residual stream x is stored as a PAIR of bf16 matrices: Xh = bf16(x), Xl = bf16(x - Xh) (17
// significant bits; Xh alone is the A operand of the GEMM that consumes the following LayerNorm). The
// three epilogues that produce x (X-epilogues) also emit, per row and per 32-column chunk, the partial
// statistics (sum, sum of squared deviations from the chunk mean) of the float32 value before the split:
// rowstat_finalize turns them into the (rstd, -mean * rstd) the consuming GEMM's epilogue applies.
enum GemmEpilogue {
    EPI_F32_POS = 0,   // x = (acc + bias) + pos[m % pos_rows] -> Xh, Xl, cstat   (patch embed)
    EPI_RESID = 1,     // x = (acc + bias) + (Xh + Xl)         -> Xh, Xl, cstat   (proj, fc2)
    EPI_GELU_BF16 = 2, // Cb bf16 = gelu(y)                                        (fc1)
    EPI_RELU_BF16 = 3, // Cb bf16 = relu(y)                                        (head convs)
    EPI_QKV = 4,       // q*ATT_Q_SCALE,k -> qk[M][2D] bf16; v -> Vt[b][h][64][npad]
    EPI_F32 = 5        // x = acc + bias                       -> Xh, Xl, cstat   (operator tests)
};
// y of the bf16 epilogues: acc + bias, or with a folded LayerNorm (rowstat != null)
//     y[m][n] = rowstat[m].x * acc + (rowstat[m].y * colsum[n] + bias[n])
// where the weights are W' = bf16(gamma * W), colsum[n] = sum_k W'[n][k], bias[n] = sum_k beta[k] W[n][k] + b[n]
#define VT_STAT_CHUNK 32

struct GemmArgs {
    const bf16_t* A; int lda;     // [M][K] row-major bf16
    const bf16_t* W; int ldw;     // [N][K] row-major bf16 (one output feature per row)
    const float* bias;            // [N]
    int M, N, K;
    bf16_t* Xh; uint8_t* Xl; int ldx;  // X-epilogues: the residual stream pair [M][ldx] - bf16 high halves, lo8 bytes (EPI_RESID: read, then written)
    float2* cstat;                // X-epilogues: [M][N / 32] chunk partials (sum, M2), or null
    // X-epilogues of the 256x256 kernel (launch_gemm reports whether it was used): the last workgroup of
    // every 256-row panel finalizes the panel's row terms itself (no launch_rowstat_finalize needed)
    float2* rowstat_out;          // [M] (rstd, -mean * rstd), or null
    unsigned* panel_cnt;          // [ceil(M / 256)] arrival counters, zero between launches
    float ln_eps;
    bf16_t* Cb; int ldcb;         // bf16 output
    const float* pos; int pos_rows;     // EPI_F32_POS: [pos_rows][ldx] f32
    const float2* rowstat;        // bf16 epilogues: folded LayerNorm row terms [M] (rstd, -mean * rstd), or null
    const float* colsum;          // ... and its column sums [N]
    const float2* cstat_in;       // 4-wave kernel only, instead of rowstat: the chunk partials [M][K / 32] of the
                                  // X-epilogue that produced A; the epilogue combines them itself (uses ln_eps)
    bf16_t* qk; bf16_t* vt; int tokens; int npad; int D;
    int vt_perm;                  // Vt key order inside each group of 16: 0 natural, 1 attn_perm16 (attention mode 3)
    unsigned long long* dbg;      // diagnostic builds only (VT_STAMPS): per-wave cycle sums
    // Implicit 3x3 convolution (zero padding) on the head's feature maps, 4-wave kernel with the ReLU
    // epilogue only: conv_grid > 0 makes A the map t[M = B*grid*grid][conv_C] (lda = conv_C) and the
    // GEMM's K = 9*conv_C the im2col row (ky*3+kx)*conv_C + c, gathered tap by tap while the LDS-DMA
    // pieces are issued (a K-tile of 64 lies inside one tap: conv_C % 64 == 0). Out-of-map taps read
    // `zeros` (>= 128 B of zeros in device memory).
    int conv_grid, conv_C;
    const bf16_t* zeros;
    // kernel of k_gemm.hip: which tiles an XCD's contiguous run of workgroups covers - 0: the launcher decides (bf16-output
    // epilogues with M < N: column tiles), 1: row panels x all columns, 2: column tiles x all rows
    int tile_order;
};

hipError_t launch_gemm(const GemmArgs& a, int epilogue, hipStream_t st);
// true: launch_gemm() will run an X-epilogue of these arguments on the 256x256 kernel, whose row panels
// finalize a.rowstat_out themselves; false: the caller launches launch_rowstat_finalize behind the GEMM
bool gemm_finalizes_rowstat(const GemmArgs& a, int epilogue);
hipError_t launch_gemm_cfg(const GemmArgs& a, int epilogue, int cfg, hipStream_t st);
int gemm_pick_config(int M, int N, int K, int epilogue, bool conv = false);
const char* gemm_config_name(int cfg);
hipError_t gemm_prepare();  // once per device, before the first launch / any stream capture
// k_gemm256.hip: the 256x256-tile 8-wave kernels (configs 18, 19); hipErrorInvalidValue = shape does not fit
#define GEMM_CFG_SMALL_MAX 8   // 0..8: the 4-wave kernel of k_gemm.hip (7, 8 with four loader waves beside it)
#define GEMM_CFG_256P4 18      // 2 phases of 32 MFMAs per K-tile, one tile per workgroup (17, round 1's 4-phase schedule: git history)
#define GEMM_CFG_256PP 19      // the same loop in persistent workgroups (bf16 outputs with more tiles than CUs; else = 18)
#define GEMM_CFG_256_MIN GEMM_CFG_256P4
// operands of the 256x256 kernels are addressed with unsigned 32-bit byte offsets from their base
#define VT_GEMM256_MAX_OPERAND_BYTES (1ll << 32)
hipError_t gemm256_prepare();
const char* gemm256_build_id();   // sha256 of k_gemm256.hip + the headers it includes + its compile flags (build.py: -DVT_TU_SHA256)
hipError_t launch_gemm256(const GemmArgs& a, int epilogue, bool persistent, hipStream_t st);
bool gemm256_fits(const GemmArgs& a, int epilogue);
// the tile configuration launch_gemm() runs for these arguments (the picker's choice, or the 4-wave
// kernel where a 256x256 choice does not fit the operands)
int gemm_effective_config(const GemmArgs& a, int epilogue);

// y[r] (bf16) = LN(x[in_row(r)]) ; in_row(r) = (r / group) * in_stride + in_off + r % group
hipError_t launch_layernorm(const float* x, const float* gamma, const float* beta, bf16_t* y,
                            int rows, int D, int group, int in_stride, int in_off, float eps,
                            hipStream_t st);
// the same on the split residual stream: x = xh + xl * 2^-12 (3-byte pair)
hipError_t launch_layernorm_split(const bf16_t* xh, const uint8_t* xl, const float* gamma, const float* beta,
                                  bf16_t* y, int rows, int D, int group, int in_stride, int in_off,
                                  float eps, hipStream_t st);
// rowstat[m] = (rstd, -mean * rstd) from the chunk partials of an X-epilogue (D = 32 * nchunk columns)
hipError_t launch_rowstat_finalize(const float2* cstat, float2* rowstat, int M, int nchunk, float eps,
                                   hipStream_t st);
// LayerNorm folded into the weights of the GEMM that consumes it (once, at engine creation):
// Wf[n][k] = bf16(gamma[k] * W[n][k]); colsum[n] = sum_k Wf[n][k]; cvec[n] = sum_k beta[k] W[n][k] + bias[n]
hipError_t launch_fold_layernorm(const bf16_t* W, const float* gamma, const float* beta, const float* bias,
                                 bf16_t* Wf, float* colsum, float* cvec, int N, int K, hipStream_t st);

// out[M][D] bf16 = softmax(q k^T) v, q/k rows in qk[M][2D], v transposed in vt[B*H][64][npad]
hipError_t launch_attention(const bf16_t* qk, const bf16_t* vt, bf16_t* out, int B, int tokens,
                            int H, int npad, hipStream_t st);

// Which attention kernel launch_attention() will run for this shape,
// and whether it wants Vt with the permuted key order (mode 3). The QKV GEMM that feeds it must be
// given the same vt_perm.
int attention_pick_mode(int tokens, int npad);
inline int attention_vt_perm(int mode) { return mode >= 3 ? 1 : 0; }
// position of key t inside its group of 16 in the permuted Vt layout: bits 2 and 3 swapped
__host__ __device__ inline int attn_perm16(int t) { return (t & ~12) | ((t & 4) << 1) | ((t & 8) >> 1); }

hipError_t attention_prepare();   // once per device, before the first launch / any stream capture
hipError_t launch_attention_mode(const bf16_t* qk, const bf16_t* vt, bf16_t* out, int B, int tokens,
                                 int H, int npad, int mode, hipStream_t st);

// crop + bilinear + normalise -> patch rows; one launch covers streams [b0, b0+nb)
// tier: the tile kernel's LDS buffer (0: 16 KiB, 1: 32 KiB, 2 or more: 64 KiB), a choice of speed only
hipError_t launch_preproc(const FrameDesc* frames, StreamState* states, bf16_t* patches,
                          const ModelDims& d, int b0, int nb, bool is_template, hipStream_t st, int tier = 0);
int preproc_tier_for_box(const ModelDims& d, float w, float h, bool is_template);

hipError_t launch_nv12_to_rgb8(const uint8_t* nv12, int w, int h, uint8_t* rgb, hipStream_t st);
// n frames per launch: d_in[i] packed NV12 (nullptr: the all-zero frame of a short buffer), d_out[i] w*h*3 bytes; the
// tables are HOST arrays of device pointers (they travel in the kernel arguments, VT_NV12_BATCH_MAX frames per launch)
#define VT_NV12_BATCH_MAX 64
struct Nv12Batch { const uint8_t* in[VT_NV12_BATCH_MAX]; uint8_t* out[VT_NV12_BATCH_MAX]; };
hipError_t launch_nv12_to_rgb8_batch(const uint8_t* const* d_in, uint8_t* const* d_out, int n, int w, int h, hipStream_t st);


hipError_t launch_overlay(uint8_t* yplane, int width, int height, int stride, const vt_draw_cmd* d_cmds,
                          int n, hipStream_t st);

hipError_t launch_overlay_rgb(uint8_t* rgb, int width, int height, int stride, const vt_draw_cmd* d_cmds,
                              int n, hipStream_t st);

// Where a pass leaves its results for the host: pinned, device-visible host memory the decode kernel
// stores to directly (no device-to-host copy on the stream). Uploaded with the frame descriptors of
// the pass, right behind them in the same buffer.
struct PassOut {
    vt_result* host_results;    // [B] or null
    StreamState* host_states;   // [B] or null
};

struct DecodeArgs {
    const bf16_t* t3;       // [B*ns][C]
    const float* w4;        // [8][C]
    const float* b4;        // [8]
    const float* hann;      // [ns]
    float* head_out;        // [B*ns][8]
    StreamState* states;    // [B]
    vt_result* results;     // [B] (device)
    const PassOut* out;     // device copy of this pass's PassOut
    int B, ns, grid, C;
    float success_threshold;
};
hipError_t launch_decode(const DecodeArgs& a, hipStream_t st);     // head_out_kernel + decode_kernel (two launches)

// The head's convolutions on the band kernel of k_head.hip: out[B*grid*grid][N] bf16 = relu(conv(in) + bias).
// conv3x3: in [B*grid*grid][C] (ldin >= C), W [N][9*C] with column (ky*3+kx)*C + c, zero padding (zeros: >= 256 B
// of zeros in device memory), N == C; else a 1x1 layer: in [B*grid*grid][K], W [N][K].
struct HeadConvArgs {
    const bf16_t* in; int ldin;
    const bf16_t* W; int ldw;
    const float* bias;
    bf16_t* out; int ldout;
    const bf16_t* zeros;
    int B, grid, C, N, K;
    int conv3x3;
    int R, ncb;                 // rows per band / 16-column blocks per wave; <= 0: planned by the launcher
    unsigned* band_cnt;         // fused tail only: [B] arrival counters, zero between launches
    float* band_best;           // fused tail only: [B][bands][2] each band's argmax candidate (response, cell), bands <= grid
    int bands, mbe_max;         // filled in by the launcher
    unsigned long long* dbg;    // diagnostic builds only (VT_STAMPS): per-wave cycle sums [wgs][8][4]
    // 1x1 layer with the final LayerNorm inside (xh != nullptr; `in` is ignored): the layer's input row (b, cell) is
    // LayerNorm(xh + xl)[b * in_stride + in_off + cell][0..K) with gamma ln_g, beta ln_b
    const bf16_t* xh;
    const uint8_t* xl;
    const float *ln_g, *ln_b;
    float ln_eps;
    int in_stride, in_off;
};
bool headconv_supported(int grid, int C, int N, int K, bool conv3x3);
bool headconv_ln_supported(int grid, int N, int D);
bool headconv_plannable(int grid, int C, int N, int K, bool conv3x3, bool tail);   // supported and some band height fits LDS
hipError_t headconv_prepare();     // once per device, before the first launch / any stream capture
// dec != nullptr (3x3 layers only): the 5-logit layer, the score window, the argmax and the box decode run inside
// the same launch (dec->t3 is ignored: the logits are computed from the layer's own output tile)
hipError_t launch_headconv(HeadConvArgs a, const DecodeArgs* dec, hipStream_t st);

// ---- kernel launches -----------------------------------------------------------------------------
// Every kernel of the library is launched through vt_launch. It is hipLaunchKernelGGL, except while the engine's
// instrumented pass (vt_group_profile_device) has armed a probe on the calling thread: the first launch behind the arming
// then goes through hipExtLaunchKernelGGL with the probe's two HIP events, which receive the begin and end timestamps of
// THAT dispatch - the duration rocprofv3's kernel trace reports - instead of the time between two marker packets around
// it (which adds the dispatch and marker latency, ~4 us per launch on this stack).
struct LaunchProbe { hipEvent_t start, stop; int launches; };
extern thread_local LaunchProbe* vt_launch_probe;
template <typename F, typename... Args>
inline void vt_launch(F kernel, const dim3& grid, const dim3& block, size_t smem, hipStream_t st, Args... args) {
    LaunchProbe* pr = vt_launch_probe;
    if (pr && pr->launches++ == 0) hipExtLaunchKernelGGL(kernel, grid, block, (uint32_t)smem, st, pr->start, pr->stop, 0, args...);
    else hipLaunchKernelGGL(kernel, grid, block, smem, st, args...);
}

// ---- small device helpers ---------------------------------------------------------------------

__device__ __forceinline__ float bf16_to_f32(bf16_t b) {
    return __uint_as_float(((uint32_t)b) << 16);
}
// f32 -> bf16, round to nearest even, NaN kept a NaN: one v_cvt_pk_bf16_f32 (gfx950). Same
// result as the integer rounding (u + 0x7fff + ((u >> 16) & 1)) >> 16 of the oracle for every
// finite input.
typedef __bf16 bf16v2_t __attribute__((ext_vector_type(2)));
typedef float f32v2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    return __builtin_bit_cast(bf16_t, (__bf16)f);
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32v2_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16v2_t));
}

// ---- one LayerNorm row by half a wave (D = 256 * NCH) -------------------------------------------
// Lane l32 of the half-wave owns the 8-float chunks c = l32 + 32 j, j < NCH. Two-pass variance; the sums run
// over a lane's chunks in ascending order, then over the 32 lanes as an xor butterfly (16, 8, 4, 2, 1). The ONE
// definition of this arithmetic: the LayerNorm kernel (k_misc.hip) and the head's first layer, which normalises
// its band's rows itself (k_head.hip), produce the same bits by construction.
// Sum over the 32 lanes of a half-wave, every lane gets the total: the xor butterfly 16, 8, 4, 2, 1 in registers -
// v_permlane16_swap (odd 16-lane rows of one operand <-> even rows of the other: both copies of v, so a lane ends up
// holding its own and its xor-16 partner's value), row_ror:8, two masked row shifts by 4 (DPP has no xor-4 pattern:
// banks 0 / 2 of a row take lane + 4, banks 1 / 3 lane - 4), two quad_perms. Same partners in the same order as
// "for (off = 16; off >= 1; off >>= 1) v += __shfl_xor(v, off)" and therefore the same bits (tools/dpp_xor_check.hip), without
// that form's five dependent LDS round trips (ds_bpermute) per sum.
// PRECONDITION - FULL EXEC: every lane of the wave must be active at the call. A DPP / permlane source lane that is
// masked off reads as 0 (bound_ctrl) or keeps the old value, so under divergence (`if (r < rows) ln_row(...)`) the sums
// would silently miss terms. All callers keep the wave uniform: rows past the end are CLAMPED to the last row and only
// the stores are guarded (k_misc.hip layernorm kernels, k_head.hip LNC path). The same holds for quad_sum / row16_sum /
// row8_sum of k_gemm_util.hpp and k_head.hip. The s_nop counts inside the asm are the gfx950 wait states (VALU write ->
// permlane read: 2; permlane write -> VALU read: 2) the compiler cannot check there; tools/dpp_xor_check.hip (run by
// tests/test_gpu_ops.py on the box, built with the box's ROCm) guards both the workaround and those counts.
__device__ __forceinline__ float half_wave_sum(float v) {
    // inline asm, not __builtin_amdgcn_permlane16_swap: on float operands hipcc (ROCm 7.2) adds the swap's FIRST result to
    // itself (the integer form compiles correctly; tools/dpp_xor_check.hip caught it). s_nop: VALU write -> permlane read
    // and permlane write -> VALU read wait states, which the compiler cannot see inside the asm.
    float lo, hi;
    asm volatile("v_mov_b32 %0, %2\n\tv_mov_b32 %1, %2\n\ts_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1"
                 : "=&v"(lo), "=&v"(hi) : "v"(v));
    v = lo + hi;
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));       // row_ror:8
    int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x104, 0xF, 0x5, false);                       // row_shl:4 -> banks 0, 2
    t = __builtin_amdgcn_update_dpp(t, __builtin_bit_cast(int, v), 0x114, 0xF, 0xA, false);                           // row_shr:4 -> banks 1, 3
    v += __builtin_bit_cast(float, t);
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));        // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));        // quad_perm [1,0,3,2]
    return v;
}

template <int NCH>
struct LnCoef { f32x4_t g[NCH][2], b[NCH][2]; };

template <int NCH>
__device__ __forceinline__ void ln_load_coef(const float* gamma, const float* beta, int l32, LnCoef<NCH>& k) {
    const f32x4_t* g4 = reinterpret_cast<const f32x4_t*>(gamma);
    const f32x4_t* b4 = reinterpret_cast<const f32x4_t*>(beta);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const int c = l32 + 32 * j;
        k.g[j][0] = g4[2 * c]; k.g[j][1] = g4[2 * c + 1];
        k.b[j][0] = b4[2 * c]; k.b[j][1] = b4[2 * c + 1];
    }
    // hipcc otherwise sinks these loads behind the row's reductions: pin the values here, all loads issued
#pragma unroll
    for (int j = 0; j < NCH; ++j)
        asm volatile("" : "+v"(k.g[j][0]), "+v"(k.g[j][1]), "+v"(k.b[j][0]), "+v"(k.b[j][1]) : : "memory");
}

// ---- the 3-byte residual pair (specification v3, top of this file) -------------------------------------------------
// decode: element k of a lane's 8 consecutive columns = bf16 half k of h (16 B) + signed byte k of l (8 B) * 2^-12
__device__ __forceinline__ float lo8_f32(uint32_t w, int byte) {
    return (float)((int)(w << (24 - 8 * byte)) >> 24);          // sign-extended byte (v_bfe_i32 / SDWA sext) -> float
}
__device__ __forceinline__ void x_join8(const u32x4_t& hi, const u32x2_t& lo, float (&x)[8]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const uint32_t lw = lo[e >> 1];
        x[2 * e] = __builtin_fmaf(lo8_f32(lw, (2 * e) & 3), VT_LO_Q, __uint_as_float(hi[e] << 16));
        x[2 * e + 1] = __builtin_fmaf(lo8_f32(lw, (2 * e + 1) & 3), VT_LO_Q, __uint_as_float(hi[e] & 0xffff0000u));
    }
}
// encode: hi = bf16(x) (round to nearest even); d = x - hi (exact); lo8 = rint(clamp(d, +-127 q) / q) by the float add of
// 1.5 * 2^11: the sum's ulp is q = 2^-12, the add rounds to nearest even, and the low byte of the sum's bit pattern
// (0x45400000 + n) is the two's-complement lo8. The add is issued in its SDWA form with dst_sel:BYTE_k, which writes that
// low byte straight into byte k of the destination dword - no extraction, no packing: 2 vector operations per element
// (v_med3_f32 + v_add_f32_sdwa) where clamp + add + v_perm gathering took 2.75 (same-box A/B in profiles/r06_lo8_residual.txt:
// + 0.5 % of the whole frame; without the clamp another + 0.2 %, not taken: a value beyond the range must saturate, not wrap).
__device__ __forceinline__ void x_split8(const float (&x)[8], u32x4_t& hi, u32x2_t& lo) {
    uint32_t h[4];
    float c[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
        c[2 * e] = __builtin_amdgcn_fmed3f(x[2 * e] - __uint_as_float(h[e] << 16), -VT_LO_MAX, VT_LO_MAX);
        c[2 * e + 1] = __builtin_amdgcn_fmed3f(x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u), -VT_LO_MAX, VT_LO_MAX);
    }
    hi = u32x4_t{h[0], h[1], h[2], h[3]};
    const float magic = 3072.0f;            // SDWA takes no literal: a register
    uint32_t w0, w1;
#define VT_LO8_BYTE(W, K, UNUSED, C) asm("v_add_f32_sdwa %0, %1, %2 dst_sel:BYTE_" #K " dst_unused:" #UNUSED " src0_sel:DWORD src1_sel:DWORD" : W : "v"(C), "v"(magic))
    VT_LO8_BYTE("=v"(w0), 0, UNUSED_PAD, c[0]); VT_LO8_BYTE("+v"(w0), 1, UNUSED_PRESERVE, c[1]);
    VT_LO8_BYTE("+v"(w0), 2, UNUSED_PRESERVE, c[2]); VT_LO8_BYTE("+v"(w0), 3, UNUSED_PRESERVE, c[3]);
    VT_LO8_BYTE("=v"(w1), 0, UNUSED_PAD, c[4]); VT_LO8_BYTE("+v"(w1), 1, UNUSED_PRESERVE, c[5]);
    VT_LO8_BYTE("+v"(w1), 2, UNUSED_PRESERVE, c[6]); VT_LO8_BYTE("+v"(w1), 3, UNUSED_PRESERVE, c[7]);
#undef VT_LO8_BYTE
    lo = u32x2_t{w0, w1};
}
// the LayerNorm readers' form: 8 values of a chunk as two float4
__device__ __forceinline__ void ln_unpack_split(const u32x4_t h, const u32x2_t l, f32x4_t (&v)[2]) {
    float x[8];
    x_join8(h, l, x);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e >> 2][e & 3] = x[e];
}

// v: the row's values of this lane (overwritten); o[j]: the normalised chunk l32 + 32 j as 8 bf16
template <int NCH>
__device__ __forceinline__ void ln_row(f32x4_t (&v)[NCH][2], const LnCoef<NCH>& k, int D, float eps, uint4 (&o)[NCH]) {
    float sum = 0.0f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int e = 0; e < 4; ++e) sum += v[j][0][e] + v[j][1][e];
    sum = half_wave_sum(sum);
    const float mean = sum / (float)D;
    float sq = 0.0f;
#pragma unroll
    for (int j = 0; j < NCH; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[j][h][e] -= mean;
                sq += v[j][h][e] * v[j][h][e];
            }
    sq = half_wave_sum(sq);
    const float rstd = 1.0f / sqrtf(sq / (float)D + eps);
#pragma unroll
    for (int j = 0; j < NCH; ++j) {
        const f32x4_t g0 = k.g[j][0], g1 = k.g[j][1], b0 = k.b[j][0], b1 = k.b[j][1];
        o[j].x = pack_bf16x2((v[j][0][0] * rstd) * g0[0] + b0[0], (v[j][0][1] * rstd) * g0[1] + b0[1]);
        o[j].y = pack_bf16x2((v[j][0][2] * rstd) * g0[2] + b0[2], (v[j][0][3] * rstd) * g0[3] + b0[3]);
        o[j].z = pack_bf16x2((v[j][1][0] * rstd) * g1[0] + b1[0], (v[j][1][1] * rstd) * g1[1] + b1[1]);
        o[j].w = pack_bf16x2((v[j][1][2] * rstd) * g1[2] + b1[2], (v[j][1][3] * rstd) * g1[3] + b1[3]);
    }
}
