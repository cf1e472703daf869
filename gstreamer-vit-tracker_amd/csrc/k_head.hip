// k_head.hip — the centre head's convolutions for gfx950: 1x1 (D -> C) and 3x3 (C -> C, zero padding)
// + bias + ReLU -> bf16, and, fused behind the last 3x3 layer, the f32 5-logit layer, the score window,
// the argmax and the box decode.
//
// Why a kernel of its own (round 5). As implicit GEMMs on the 4-wave kernel of k_gemm.hip the four head
// layers of a 30-stream pass took 93 us at 0.07-0.08 of the MFMA peak: M = 17,280, N = 128 gives 540
// tiles of 64x64 that each walk all of K = 1152 as a dependent chain of LDS-DMA round trips, and every
// tile re-fetches its A rows nine times (once per tap). What bounds such a layer is what ONE CU can pull
// through its vector-memory path (~70 GB/s from L2, MI355X guide "Indexed rows: gather into LDS"), so the
// kernel is built around bytes per CU:
//   * a workgroup owns a BAND of R rows of one stream's S x S feature map and all (or half) of the output
//     channels: 256 CUs x one band each covers a 30-stream pass in one round;
//   * 3x3 layers: the band's input cells plus a one-cell halo go to LDS ONCE ((R + 2)(S + 2) cells,
//     zero page for cells outside the map); the nine taps are nine shifted views of that image - the A
//     operand is read from L2 1.6x instead of 9x;
//   * the weights stream through a ring of K-tiles (one tap = one K-tile of C channels) filled by LDS-DMA
//     two tiles ahead, one raw barrier per tap, counted vmcnt;
//   * 8 waves: four compute (wave w owns output channels [w * BN/4, (w + 1) * BN/4) of every cell of the band:
//     D^T = W . A^T with v_mfma_f32_16x16x32_bf16, so a lane holds 4 consecutive channels of one cell), four
//     only issue the LDS-DMA pieces (their issue stalls run beside the MFMA streams, not in them);
//   * the bf16 tile is assembled in the dead ring and leaves as whole 16-B pieces of contiguous rows.
// The 1x1 layer (K = D) streams its A rows through the same ring (K-tile depth 64).
// TAIL (last 3x3 layer): the band's 5 logits per cell are computed from the staged bf16 tile (f32 weights),
// written with write-through stores; the last band of a stream to arrive (ticket on a per-stream counter,
// the guide's first valid hand-off form) reads the stream's logits with sc1 loads and decodes the box:
// head_out and decode are no launches of their own any more (7 -> 4 launches behind the final LayerNorm).
//
// Numerics: every output element is one MFMA accumulation chain over k = tap * C + c in ascending order,
// bias added in f32, ReLU, round-to-nearest-even to bf16 - bit-identical to the implicit-GEMM kernel
// (tests/test_gpu_ops.py). The decode is the float-op order of vto_decode (oracle/vt_oracle.c).
#include "vt_common.hpp"
#include "k_gemm_util.hpp"
#include <algorithm>

#define HC_MBMAX 7            // row blocks of 16 cells per band (112 cells)

// ---- score window + argmax + box decode of ONE stream (256 threads) ---------------------------------
// SC1: head_out is read with sc1 loads (handed over by other workgroups of the same launch); plain loads
// otherwise (written by an earlier launch).
template <bool SC1>
__device__ __forceinline__ void load_logits(const float* ho, float (&o)[5]) {
    if constexpr (SC1) {
        u32x4_t a; uint32_t b;
        asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dword %1, %2, off offset:16 sc1\n\ts_waitcnt vmcnt(0)"
                     : "=&v"(a), "=&v"(b) : "v"(ho) : "memory");
        o[0] = __uint_as_float(a[0]); o[1] = __uint_as_float(a[1]); o[2] = __uint_as_float(a[2]);
        o[3] = __uint_as_float(a[3]); o[4] = __uint_as_float(b);
    } else {
#pragma unroll
        for (int k = 0; k < 5; ++k) o[k] = ho[k];
    }
}

__device__ __forceinline__ float hc_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }

// better(a, b): candidate a = (response, cell) beats b: larger response, the lower cell on a tie - what "first
// maximum in ascending cell order" (vto_decode) selects, whatever the order the candidates are combined in
__device__ __forceinline__ bool hc_better(float ra, int ia, float rb, int ib) { return ra > rb || (ra == rb && ia < ib); }

// The 3x3 window around the argmax cell and the box: threads 0..8 evaluate the window's cells, thread 0 adds the
// terms in vto_decode's order (dy, dx ascending: bit-identical to the serial form) and writes the result. s_win:
// 9 x 5 floats + 1 (the argmax cell's score logit) of LDS; every thread of the block calls this (one barrier).
// pre / pre_po (thread 0 only, may be null): the stream's state and the pass's output addresses as fetched earlier
// (nothing else writes them during the launch).
template <bool SC1, bool PRE = false>
__device__ __forceinline__ void decode_box(const DecodeArgs& a, int b, int idx, int tid, float* s_win,
                                           const StreamState& pre = StreamState{}, const PassOut& pre_po = PassOut{}) {
    const int grid = a.grid, ns = a.ns;
    const int bx = idx % grid, by = idx / grid;
    if (tid < 9) {
        const int ix = bx + tid % 3 - 1, iy = by + tid / 3 - 1;
        float t[5] = {-1.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        if (ix >= 0 && iy >= 0 && ix < grid && iy < grid) {
            float o[5];
            const float hv = a.hann[iy * grid + ix];       // issued ahead of the logits' loads, not behind their wait
            load_logits<SC1>(a.head_out + ((size_t)b * ns + iy * grid + ix) * 8, o);
            const float r = hc_sigmoid(o[0]) * hv;
            const float w = r * r;
            const float offx = 3.0f * hc_sigmoid(o[1]) - 1.0f;
            const float offy = 3.0f * hc_sigmoid(o[2]) - 1.0f;
            const float cxj = ((float)ix + offx) / (float)grid;
            const float cyj = ((float)iy + offy) / (float)grid;
            t[0] = w; t[1] = w * cxj; t[2] = w * cyj; t[3] = w * hc_sigmoid(o[3]); t[4] = w * hc_sigmoid(o[4]);
            if (tid == 4) s_win[45] = o[0];       // the window's centre is the argmax cell
        } else if (tid == 4) {
            s_win[45] = __builtin_nanf("");       // no cell had a comparable response (NaN logits): score NaN, success 0
        }
#pragma unroll
        for (int k = 0; k < 5; ++k) s_win[tid * 5 + k] = t[k];
    }
    __syncthreads();
    if (tid != 0) return;
    StreamState s;
    if constexpr (PRE) s = pre; else s = a.states[b];
    const float score = hc_sigmoid(s_win[45]);
    float sw = 0.0f, scx = 0.0f, scy = 0.0f, sbw = 0.0f, sbh = 0.0f;
    for (int j = 0; j < 9; ++j) {
        if (s_win[j * 5] < 0.0f) continue;
        sw = sw + s_win[j * 5];
        scx = scx + s_win[j * 5 + 1];
        scy = scy + s_win[j * 5 + 2];
        sbw = sbw + s_win[j * 5 + 3];
        sbh = sbh + s_win[j * 5 + 4];
    }
    const float cxn = scx / sw, cyn = scy / sw, wn = sbw / sw, hn = sbh / sw;
    const float side = s.geo[3];
    const float cx = (s.geo[0] + 0.5f) + cxn * side;
    const float cy = (s.geo[1] + 0.5f) + cyn * side;
    float bw = wn * side, bh = hn * side;
    float x1 = cx - 0.5f * bw, y1 = cy - 0.5f * bh;
    float x2 = x1 + bw, y2 = y1 + bh;
    const float margin = 10.0f;
    const float W = (float)s.frame_w, Hh = (float)s.frame_h;
    x1 = fminf(fmaxf(0.0f, x1), W - margin);
    y1 = fminf(fmaxf(0.0f, y1), Hh - margin);
    x2 = fminf(fmaxf(margin, x2), W);
    y2 = fminf(fmaxf(margin, y2), Hh);
    bw = fmaxf(margin, x2 - x1);
    bh = fmaxf(margin, y2 - y1);
    const int success = (score >= a.success_threshold) ? 1 : 0;
    vt_result r;
    r.success = success;
    r.score = score;
    r.bbox.x = (int32_t)floorf(x1 + 0.5f);
    r.bbox.y = (int32_t)floorf(y1 + 0.5f);
    r.bbox.width = (int32_t)floorf(bw + 0.5f);
    r.bbox.height = (int32_t)floorf(bh + 0.5f);
    a.results[b] = r;
    s.last_fbox[0] = x1; s.last_fbox[1] = y1; s.last_fbox[2] = bw; s.last_fbox[3] = bh;
    s.last_score = score;
    s.last_idx = idx;
    s.frames_done += 1;
    if (success) {
        // The state the next frame's crop is cut around is the INTEGER box the caller sees (≙ the
        // reference's BBox{i32}): two implementations whose boxes agree then cut bit-identical crops
        // and re-synchronise exactly.
        s.success_count += 1;
        s.box[0] = (float)r.bbox.x; s.box[1] = (float)r.bbox.y;
        s.box[2] = (float)r.bbox.width; s.box[3] = (float)r.bbox.height;
    }
    // the host's copies, straight into its pinned memory (visible to it once the pass's event or the
    // stream synchronises; the fence orders the stores ahead of the kernel's end for every scope)
    PassOut po;
    if constexpr (PRE) po = pre_po; else po = *a.out;
    if (po.host_results) po.host_results[b] = r;
    if (po.host_states) po.host_states[b] = s;
    a.states[b] = s;
    __threadfence_system();
}

// the whole stream by one block (the decode as a launch of its own): argmax over all cells, then decode_box.
// smem: >= 2048 + 46 * 4 bytes, 4-B aligned. Same float op order as vto_decode (oracle/vt_oracle.c).
template <bool SC1>
__device__ __forceinline__ void decode_stream(const DecodeArgs& a, int b, int tid, char* smem) {
    float* s_best = reinterpret_cast<float*>(smem);
    int* s_idx = reinterpret_cast<int*>(smem + 1024);
    float* s_win = reinterpret_cast<float*>(smem + 2048);
    const int ns = a.ns;
    float best = -1.0f;
    int bidx = 0x7fffffff;
    for (int i = tid; i < ns; i += 256) {
        float o[5];
        load_logits<SC1>(a.head_out + ((size_t)b * ns + i) * 8, o);
        const float resp = hc_sigmoid(o[0]) * a.hann[i];
        if (hc_better(resp, i, best, bidx)) { best = resp; bidx = i; }
    }
    s_best[tid] = best;
    s_idx[tid] = bidx;
    __syncthreads();
    for (int off = 128; off >= 1; off >>= 1) {
        if (tid < off && hc_better(s_best[tid + off], s_idx[tid + off], s_best[tid], s_idx[tid])) {
            s_best[tid] = s_best[tid + off];
            s_idx[tid] = s_idx[tid + off];
        }
        __syncthreads();
    }
    const int idx = s_idx[0];
    __syncthreads();
    decode_box<SC1>(a, b, idx, tid, s_win);
}

// one block per stream: the decode as a launch of its own (head_out written by an earlier launch)
__global__ __launch_bounds__(256) void decode_kernel(DecodeArgs a) {
    __shared__ __attribute__((aligned(16))) char smem[2048 + 48 * 4];
    decode_stream<false>(a, blockIdx.x, threadIdx.x, smem);
}

// ---- last head layer (C -> 5 logits, f32): one wave per search token (kept for the engines whose last
// 3x3 layer runs on the implicit-GEMM kernel, and for the operator tests) ---------------------------
__global__ __launch_bounds__(256) void head_out_kernel(const bf16_t* __restrict__ t3, const float* __restrict__ w4,
                                                       const float* __restrict__ b4, float* __restrict__ head_out,
                                                       int rows, int C) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const bf16_t* row = t3 + (size_t)r * C;
    float o[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int c = lane * 2; c < C; c += 128) {
        const uint32_t pk = *reinterpret_cast<const uint32_t*>(row + c);
        const float t0 = __uint_as_float(pk << 16), t1 = __uint_as_float(pk & 0xffff0000u);
#pragma unroll
        for (int k = 0; k < 5; ++k) o[k] += t0 * w4[k * C + c] + t1 * w4[k * C + c + 1];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) o[k] += __shfl_xor(o[k], off);
    if (lane == 0) {
        float* ho = head_out + (size_t)r * 8;
#pragma unroll
        for (int k = 0; k < 5; ++k) ho[k] = o[k] + b4[k];
        ho[5] = ho[6] = ho[7] = 0.0f;
    }
}

hipError_t launch_decode(const DecodeArgs& a, hipStream_t st) {
    if (a.C % 2 != 0) return hipErrorInvalidValue;
    const int rows = a.B * a.ns;
    vt_launch(head_out_kernel, dim3((rows + 3) / 4), dim3(256), 0, st, a.t3, a.w4, a.b4, a.head_out,
                       rows, a.C);
    vt_launch(decode_kernel, dim3(a.B), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ---- the band kernel -----------------------------------------------------------------------------------
// HALO: 3x3 layer (A image resident, K-tile = one tap of BK = C channels); else 1x1 layer (A rows stream
// through the ring, K-tile depth BK = 64). NCB: 16-column blocks per wave (BN = 64 * NCB output channels per
// workgroup). MB: row blocks of 16 cells per band, a compile-time constant so that the main loop is straight-line
// code with its LDS reads hoisted over the MFMAs (with a run-time count hipcc split the loop body into one basic
// block per row block, each waiting for its own read); a short last band computes its padding blocks on the
// band's last cell and stores nothing for them. TAIL: + logits + decode (needs BN = N = C).
// sum over the 16 lanes of a DPP row (lanes 16k .. 16k+15), every lane gets the same bits: two quad_perm moves,
// row_half_mirror, row_mirror - no LDS round trip (a ds_bpermute butterfly was 20 dependent LDS trips per cell group)
// (full EXEC required for row16_sum / row8_sum: see half_wave_sum in vt_common.hpp)
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));
    return v;
}
// the same over 8 lanes (lanes 8k .. 8k+7)
__device__ __forceinline__ float row8_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));
    return v;
}

// 512 threads: waves 0-3 compute (one per SIMD; wave w owns output channels [w * BN/4, (w + 1) * BN/4) of every cell of
// the band), waves 4-7 are LOADERS: every LDS-DMA of the kernel is theirs. One global_load_lds costs the issuing wave ~100
// cycles of back-pressure from the CU's vector-memory path (a tap's 8 pieces per wave = longer than its 40 MFMAs); issued
// by the computing waves themselves (first version) the two added up - 12.3 us per 30-stream layer, of which the loop was
// issue stalls; a loader wave stalls beside a computing wave's MFMA stream instead (the guide's ring-gemm structure).
// One raw barrier per K-tile for all eight waves: the loaders pass it after their counted vmcnt (tile kt has landed),
// the computing waves after the MFMAs of tile kt - 1 (its stage may be refilled).
// LNC > 0 (1x1 layer only): the layer's input is the final LayerNorm of the split residual stream (D = 256 * LNC),
// computed by the workgroup itself for its band's rows: all eight waves normalise the rows (half a wave per row, the
// arithmetic of ln_row - the LayerNorm kernel's own) straight into a RESIDENT A image [K-tile][row][128 B] in the ring's
// swizzle, while the first weight tiles are in flight; the main loop then streams the weights only. The LayerNorm as a
// launch of its own (53 MB read, 26.5 MB written, and the 1x1 layer reading them back: 17 + 12 us of a 30-stream pass at
// cfg3) becomes the 53 MB read of this kernel.
template <int BK, int NCB, int MB, bool HALO, bool TAIL, int LNC = 0>
__global__ __launch_bounds__(512) void head_conv_kernel(HeadConvArgs p, DecodeArgs dec) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NT = 512;
    constexpr int ROWB = BK * 2;                  // bytes per LDS row: one cell's channels of a tap / one K-tile row
    constexpr int CPR = ROWB / 16;                // 16-B chunks per row (8 or 16)
    constexpr int RPP = 1024 / ROWB;              // rows per 1-KiB LDS-DMA piece (8 or 4)
    constexpr int BN = 64 * NCB;
    constexpr bool RES = HALO || LNC > 0;         // the A operand is resident in LDS: the ring carries weights only
    constexpr int NS = RES ? 3 : 4;               // ring stages
    constexpr int WPW = BN / RPP / 4;             // W pieces per loader wave and stage
    static_assert(BK == 64 || BK == 128, "K-tile depth");
    static_assert(!TAIL || HALO, "the fused tail follows a 3x3 layer");
    static_assert(LNC == 0 || (!HALO && BK == 64), "the fused LayerNorm feeds the 1x1 layer");
    // LDS images: rows of ROWB bytes, 16-B chunk c of a row stored at chunk slot slot_of(c, key) (LDS-DMA writes lane-
    // linear, so the permutation is applied to the per-lane SOURCE chunk - chunk_at - and again on the read).
    // 128-B rows (two per bank line): slot = c ^ key with key = row >> 1 - the 4-wave GEMM's swizzle, conflict-free for the
    // ds_read_b128 lane groups when a group's 16 rows start at a multiple of 16.
    // 256-B rows (one bank line each): the lane groups of ds_read_b128 ({0-3, 12-15, 20-27}, ...) mix two k-chunks - 8
    // lanes read chunk c, 8 read chunk c ^ 1 - of 16 CONSECUTIVE rows that start ANYWHERE here (the 3x3 taps shift the
    // window cell by cell). slot = (c & 1) * 8 + (((c >> 1) ^ key) & 7): the chunk's low bit picks the half of the line,
    // so the two sets of lanes never meet, and within a half 8 consecutive keys give 8 distinct slots whatever the start
    // (the first version, c ^ (row & 15), was 2-way on every odd start and at every wrap of a map row: PMC 31 % of the
    // LDS cycles of a 3x3 layer were bank conflicts). key of an A-image cell = its UNPADDED map index, which stays
    // consecutive across the ends of the map's rows; of a weight row its index.
    auto slot_of = [](int c, int key) { return BK == 64 ? (c ^ key) & 7 : ((c & 1) << 3) | (((c >> 1) ^ key) & 7); };
    auto chunk_at = [](int slot, int key) { return BK == 64 ? (slot ^ key) & 7 : ((((slot & 7) ^ key) & 7) << 1) | (slot >> 3); };
    auto key_of_row = [](int row) { return BK == 64 ? row >> 1 : row; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave >= 4;                // wave-uniform
    const int w4 = wave & 3;
    const int l15 = lane & 15, lq = lane >> 4;

    // which band: blockIdx.x = (stream * bands + band) * ncol + column group
    const int ncol = p.N / BN;
    const int cg = blockIdx.x % ncol, bb = blockIdx.x / ncol;
    const int band = bb % p.bands, b = bb / p.bands;
    const int grid = p.grid, ns = grid * grid;
    const int y0 = band * p.R;
    const int rows_eff = min(p.R, grid - y0);
    const int cells = rows_eff * grid;            // output cells of this band (<= 16 * MB)
    const int m_base = b * ns + y0 * grid;        // first output row of the band in [B * ns][N]
    const int n0 = cg * BN;

    // LDS: [A image (HALO) | ring of NS stages]; a stage = [A rows (stream mode) | BN weight rows]
    const int hw = grid + 2;
    const int hcells = HALO ? (p.R + 2) * hw : 0;
    const int a_tile = LNC > 0 ? p.R * grid * ROWB : 0;        // LNC: one K-tile of the resident A image (all band rows)
    const int himg = HALO ? ((hcells + RPP - 1) / RPP) * RPP * ROWB : LNC > 0 ? ((p.K / BK) * a_tile + 1023) & ~1023 : 0;
    constexpr int MBE = (MB + 1) & ~1;            // stream mode: an even number of row blocks (pieces divide over 4 waves)
    constexpr int a_stage = RES ? 0 : MBE * 16 * ROWB;
    constexpr int stage = a_stage + BN * ROWB;
    constexpr int APW = RES ? 0 : MBE / 2;         // stream mode: MBE / 2 pieces of 8 rows per loader wave (ROWB 128)
    constexpr int IPS = WPW + APW;                 // LDS-DMA pieces per loader wave and stage
    char* ring = smem + himg;
    const int ntile = HALO ? 9 : p.K / BK;

    // the fused tail's operands that do not depend on the layer's result are fetched BEFORE the main loop (plain loads,
    // consumed behind it): the logit weights of this lane's 8 channels, the Hann values of its cells, and - thread 0 -
    // the stream state and the pass's output addresses. Behind the loop each was a dependent round trip at the very
    // end of the pass.
    constexpr int LPC = CPR;                      // TAIL: lanes per cell (8 channels each)
    constexpr int CPI = NT / LPC;                 // cells per iteration of the logits loop
    constexpr int NIT = (MB * 16 + CPI - 1) / CPI;
    const int sub = tid % LPC;
    float w4r[5][8], hannr[NIT], b4r[5];
    if constexpr (TAIL) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const f32x4_t lo = *reinterpret_cast<const f32x4_t*>(dec.w4 + k * BN + sub * 8);
            const f32x4_t hi = *reinterpret_cast<const f32x4_t*>(dec.w4 + k * BN + sub * 8 + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) { w4r[k][e] = lo[e]; w4r[k][4 + e] = hi[e]; }
            b4r[k] = dec.b4[k];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cell = it * CPI + tid / LPC;
            hannr[it] = dec.hann[y0 * grid + (cell < cells ? cell : cells - 1)];
        }
    }

    // LNC: the band's rows, normalised into the A image. Half-wave h of the 16 takes rows h, h + 16, ...; the raw rows
    // are fetched two iterations ahead (a row is one memory round trip, a band five of them back to back otherwise).
    auto ln_phase = [&]() {
        if constexpr (LNC > 0) {
            const int l32 = lane & 31, hwv = tid >> 5;
            const int D = p.K;
            const size_t tok0 = (size_t)b * p.in_stride + p.in_off + y0 * grid;
            u32x4_t rh[3][LNC];
            u32x2_t rl[3][LNC];            // the row's lo8 bytes (3-byte residual pair, vt_common.hpp)
            auto fetch = [&](int it, int slot) {
                int r = it * 16 + hwv;
                r = r < cells ? r : cells - 1;
                const u32x4_t* hr = reinterpret_cast<const u32x4_t*>(p.xh + (tok0 + r) * D);
                const u32x2_t* lr = reinterpret_cast<const u32x2_t*>(p.xl + (tok0 + r) * D);
#pragma unroll
                for (int j = 0; j < LNC; ++j) { rh[slot][j] = hr[l32 + 32 * j]; rl[slot][j] = lr[l32 + 32 * j]; }
            };
            fetch(0, 0);
            if (MB > 1) fetch(1, 1);
            LnCoef<LNC> k;
            ln_load_coef<LNC>(p.ln_g, p.ln_b, l32, k);
#pragma unroll
            for (int it = 0; it < MB; ++it) {
                if (it + 2 < MB) fetch(it + 2, (it + 2) % 3);
                const int r = it * 16 + hwv;
                f32x4_t v[LNC][2];
#pragma unroll
                for (int j = 0; j < LNC; ++j) ln_unpack_split(rh[it % 3][j], rl[it % 3][j], v[j]);
                uint4 o[LNC];
                ln_row<LNC>(v, k, D, p.ln_eps, o);
                if (r < cells) {
#pragma unroll
                    for (int j = 0; j < LNC; ++j) {
                        const int c = l32 + 32 * j;                   // 16-B chunk of the row: K-tile c >> 3, chunk c & 7
                        *reinterpret_cast<uint4*>(smem + (c >> 3) * a_tile + r * ROWB + (slot_of(c & 7, key_of_row(r)) << 4)) = o[j];
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the rows are in LDS before this wave's first barrier
        }
    };

    f32x4_t acc[MB][NCB];

    if (loader) {
        // ---- loader waves: A image, then the ring -------------------------------------------------------------
        if constexpr (HALO) {
            const int npieces = himg / 1024;
            for (int pc = w4; pc < npieces; pc += 4) {
                const int h = pc * RPP + lane / CPR;                 // halo cell of this lane
                const int hy = h / hw, hx = h - hy * hw;
                const int y = y0 + hy - 1, x = hx - 1;
                const int key = BK == 64 ? h >> 1 : y * grid + x;     // 256-B rows: the cell's unpadded map index
                const int c = chunk_at(lane % CPR, key);             // global chunk that lands in LDS chunk slot lane % CPR
                const bool in = h < hcells && y >= 0 && y < grid && x >= 0 && x < grid;
                const bf16_t* src = in ? p.in + ((size_t)(b * ns + y * grid + x)) * p.ldin + c * 8 : p.zeros + c * 8;
                glds16(src, smem + pc * 1024);
            }
        }
        // per-lane source pointers of the pieces this wave issues per stage
        const bf16_t* wsrc[WPW];
#pragma unroll
        for (int j = 0; j < WPW; ++j) {
            const int row = (w4 * WPW + j) * RPP + lane / CPR;
            const int c = chunk_at(lane % CPR, key_of_row(row));
            wsrc[j] = p.W + (size_t)(n0 + row) * p.ldw + c * 8;
        }
        const bf16_t* asrc[APW + 1];
        if constexpr (!HALO) {
#pragma unroll
            for (int j = 0; j < APW; ++j) {
                const int row = (w4 * APW + j) * RPP + lane / CPR;
                const int c = chunk_at(lane % CPR, key_of_row(row));
                const int rc = row < cells ? row : cells - 1;        // padding rows repeat the last cell, never stored
                asrc[j] = p.in + (size_t)(m_base + rc) * p.ldin + c * 8;
            }
        }
        auto issue_tile = [&](int kt, int buf) {
            char* st = ring + buf * stage;
#pragma unroll
            for (int q = 0; q < APW; ++q) glds16(asrc[q] + kt * BK, st + (w4 * APW + q) * 1024);
#pragma unroll
            for (int q = 0; q < WPW; ++q) glds16(wsrc[q] + kt * BK, st + a_stage + (w4 * WPW + q) * 1024);
        };
#pragma unroll
        for (int s_ = 0; s_ < NS - 1; ++s_)
            if (s_ < ntile) issue_tile(s_, s_);
        ln_phase();                                  // (LNC) beside the first weight tiles' flight
        int cur = 0;
#ifdef VT_STAMPS
        unsigned long long st_wait = 0, st_bar = 0, st_issue = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a, st_b;
#define HSTAMP(v) { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define HSTAMP(v)
#endif
        for (int kt = 0; kt < ntile; ++kt) {
            HSTAMP(st_a)
            const int ahead = min(NS - 2, ntile - 1 - kt);       // tiles that may stay in flight behind tile kt
            if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * IPS>();
            else if (ahead >= 1) wait_vmcnt<IPS>();
            else wait_vmcnt<0>();
#ifdef VT_STAMPS
            HSTAMP(st_b) st_wait += st_b - st_a;
#endif
            __builtin_amdgcn_s_barrier();           // tile kt has landed; the computing waves are done with tile kt - 1
            __builtin_amdgcn_sched_barrier(0);
#ifdef VT_STAMPS
            HSTAMP(st_a) st_bar += st_a - st_b;
#endif
            if (kt + NS - 1 < ntile) issue_tile(kt + NS - 1, cur == 0 ? NS - 1 : cur - 1);
#ifdef VT_STAMPS
            HSTAMP(st_b) st_issue += st_b - st_a;
#endif
            cur = (cur + 1 == NS) ? 0 : cur + 1;
        }
#ifdef VT_STAMPS
        if (p.dbg && lane == 0) {
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
            d[0] = st_wait; d[1] = st_bar; d[2] = st_issue; d[3] = __builtin_amdgcn_s_memtime() - st_t0;
        }
#endif
    } else {
        // ---- computing waves -------------------------------------------------------------------------------------
        ln_phase();
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NCB; ++j) acc[i][j] = f32x4_t{0.0f, 0.0f, 0.0f, 0.0f};
        // fragment addresses. W rows of this wave: wave * 16 * NCB + j * 16 + l15
        int woff[NCB], wsw[NCB];
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int row = w4 * 16 * NCB + j * 16 + l15;
            woff[j] = a_stage + row * ROWB;
            wsw[j] = key_of_row(row);
        }
        // A rows: cell i * 16 + l15 of the band (padding rows use the band's last cell)
        int h0[MB];                                    // HALO: halo index of the cell's tap (0, 0); else LDS row
        int cm[MB];                                    // HALO: unpadded map index of the cell's tap (0, 0) (swizzle key)
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            int cell = i * 16 + l15;
            cell = cell < cells ? cell : cells - 1;
            if constexpr (HALO) {
                const int y = cell / grid, x = cell - y * grid;
                h0[i] = y * hw + x;
                cm[i] = (y0 + y - 1) * grid + x - 1;
            } else {
                h0[i] = LNC > 0 ? cell : i * 16 + l15;     // resident A image: the band's rows only
                cm[i] = 0;
            }
        }
        int cur = 0;
#ifdef VT_STAMPS
        unsigned long long st_bar = 0, st_comp = 0, st_t0 = __builtin_amdgcn_s_memtime(), st_a, st_b;
#endif
        for (int kt = 0; kt < ntile; ++kt) {
            HSTAMP(st_a)
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef VT_STAMPS
            HSTAMP(st_b) st_bar += st_b - st_a;
#endif
            // fragment addresses of the tap's first k-step: LDS byte offset of chunk lq of each row. The chunk of k-step
            // ks is 4 ks + lq, and in both swizzles that moves the slot by an XOR of bits that lq and the key do not carry
            // (256-B rows: slot ^ 2 ks, 128-B rows: slot ^ 4 ks): one v_xor per address and k-step instead of the whole
            // swizzle (3 VALU ops each - with one wave per SIMD they are issued between the MFMAs, not beside them: 16 of
            // them and 7 LDS reads per 10 MFMAs made a k-step 333 cycles for 160 of matrix pipe).
            const uint32_t sbase = (uint32_t)(himg + cur * stage);           // this stage, as an offset into smem
            uint32_t a0[MB], w0[NCB];
            if constexpr (HALO) {
                const int ky = kt / 3, kx = kt - 3 * ky;
                const int toff = ky * hw + kx, koff = ky * grid + kx;
#pragma unroll
                for (int i = 0; i < MB; ++i) {
                    const int h = h0[i] + toff;
                    a0[i] = (uint32_t)(h * ROWB + (slot_of(lq, BK == 64 ? h >> 1 : cm[i] + koff) << 4));
                }
            } else {
                const uint32_t abase = LNC > 0 ? (uint32_t)(kt * a_tile) : sbase;
#pragma unroll
                for (int i = 0; i < MB; ++i) a0[i] = abase + (uint32_t)(h0[i] * ROWB + (slot_of(lq, key_of_row(h0[i])) << 4));
            }
#pragma unroll
            for (int j = 0; j < NCB; ++j) w0[j] = sbase + (uint32_t)(woff[j] + (slot_of(lq, wsw[j]) << 4));
            constexpr int KS = BK / 32;
            constexpr uint32_t KSX = BK == 64 ? 64u : 32u;       // byte-offset XOR per k-step
            // k-steps software-pipelined inside the wave (one computing wave per SIMD: nobody else hides an LDS round
            // trip): the fragments of step ks + 1 are read while the MFMAs of step ks issue
            bf16x8_t wf[2][NCB], af[2][MB];
#pragma unroll
            for (int j = 0; j < NCB; ++j) wf[0][j] = *reinterpret_cast<const bf16x8_t*>(smem + w0[j]);
#pragma unroll
            for (int i = 0; i < MB; ++i) af[0][i] = *reinterpret_cast<const bf16x8_t*>(smem + a0[i]);
            // The read block and the MFMA block of a k-step are fenced for the scheduler: left alone hipcc interleaved the
            // reads among the MFMAs with an lgkmcnt(0) in front of every pair of MFMAs - each wait exposed a full LDS round
            // trip to the one wave of the SIMD (in-kernel stamps: 13.9 k cycles of compute for 5.8 k cycles of MFMAs per
            // band, the loaders idle at the barrier). Fenced, the reads of step ks + 1 are all in flight before the first
            // MFMA of step ks, which waits with a counted lgkmcnt for step ks's fragments only.
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int cb = ks & 1, nb = cb ^ 1;
                if (ks + 1 < KS) {
#pragma unroll
                    for (int j = 0; j < NCB; ++j)
                        wf[nb][j] = *reinterpret_cast<const bf16x8_t*>(smem + (w0[j] ^ ((ks + 1) * KSX)));
#pragma unroll
                    for (int i = 0; i < MB; ++i)
                        af[nb][i] = *reinterpret_cast<const bf16x8_t*>(smem + (a0[i] ^ ((ks + 1) * KSX)));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int j = 0; j < NCB; ++j)   // D[n][m]: rows = output channels (registers), columns = cells (lanes)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[cb][j], af[cb][i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef VT_STAMPS
            HSTAMP(st_a) st_comp += st_a - st_b;
#endif
            cur = (cur + 1 == NS) ? 0 : cur + 1;
        }
#ifdef VT_STAMPS
        if (p.dbg && lane == 0) {
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
            d[0] = 0; d[1] = st_bar; d[2] = st_comp; d[3] = __builtin_amdgcn_s_memtime() - st_t0;
        }
#endif
    }
    __syncthreads();                               // every wave is done with the ring (no LDS-DMA outstanding)
    // TAIL: the stream's state and the pass's output addresses, for the workgroup that will turn out to be the stream's
    // last: fetched here by thread 0 with inline-asm loads (hipcc would sink plain loads to their use - a dependent round
    // trip at the very end of the pass - or, hoisted above the main loop, spill them around it), consumed behind the
    // ticket; nothing else writes them during this launch.
    uint2 pre_s[11], pre_o[2];                     // 8-B loads: both structs are 8-B aligned in their arrays
    static_assert(sizeof(StreamState) == 88 && sizeof(PassOut) == 16, "prefetch layout");
    if constexpr (TAIL) {
        if (tid == 0) {
            const char* sp = reinterpret_cast<const char*>(dec.states + b);
            const char* op = reinterpret_cast<const char*>(dec.out);
#pragma unroll
            for (int i = 0; i < 11; ++i) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre_s[i]) : "v"(sp + 8 * i) : "memory");
#pragma unroll
            for (int i = 0; i < 2; ++i) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(pre_o[i]) : "v"(op + 8 * i) : "memory");
        }
    }

    // ---- epilogue: bias + ReLU -> bf16 tile [cells][BN] in the dead ring -> whole rows out ------------------
    constexpr int OSTR = BN * 2 + 16;              // padded row: 8-B writes of 16 lanes in 16 rows hit distinct banks
    char* otile = ring;
    if (!loader) {
#pragma unroll
        for (int j = 0; j < NCB; ++j) {
            const int nl = w4 * 16 * NCB + j * 16 + 4 * lq;          // this lane's 4 consecutive channels
            const f32x4_t bias = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + nl);
#pragma unroll
            for (int i = 0; i < MB; ++i) {
                const int m = i * 16 + l15;
                const float v0 = fmaxf(acc[i][j][0] + bias[0], 0.0f), v1 = fmaxf(acc[i][j][1] + bias[1], 0.0f);
                const float v2 = fmaxf(acc[i][j][2] + bias[2], 0.0f), v3 = fmaxf(acc[i][j][3] + bias[3], 0.0f);
                *reinterpret_cast<uint2*>(otile + m * OSTR + nl * 2) = make_uint2(pack_bf16x2(v0, v1), pack_bf16x2(v2, v3));
            }
        }
    }
    __syncthreads();
    {
        constexpr int CH = BN / 8;                 // 16-B pieces per row
        for (int c = tid; c < cells * CH; c += NT) {
            const int r = c / CH, ch = c - r * CH;
            *reinterpret_cast<uint4*>(p.out + (size_t)(m_base + r) * p.ldout + n0 + ch * 8) =
                *reinterpret_cast<const uint4*>(otile + r * OSTR + ch * 16);
        }
    }
    if constexpr (TAIL) {
        // ---- 5 logits per cell from the staged tile: LPC lanes x 8 channels per cell, f32 weights ------------------
        // (BN = N = C = BK here: a row is CPR chunks of 8 channels). The band's own argmax candidate: response =
        // sigmoid(score logit) * hann, kept in the (dead) A image.
        float* s_resp = reinterpret_cast<float*>(smem + 64);          // [MB * 16]
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int cell = it * CPI + tid / LPC;
            const int cc = cell < cells ? cell : cells - 1;
            const uint4 v = *reinterpret_cast<const uint4*>(otile + cc * OSTR + sub * 16);
            const uint32_t u[4] = {v.x, v.y, v.z, v.w};
            float o[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t0 = __uint_as_float(u[e] << 16), t1 = __uint_as_float(u[e] & 0xffff0000u);
#pragma unroll
                for (int k = 0; k < 5; ++k) o[k] += t0 * w4r[k][2 * e] + t1 * w4r[k][2 * e + 1];
            }
#pragma unroll
            for (int k = 0; k < 5; ++k) o[k] = (LPC == 16 ? row16_sum(o[k]) : row8_sum(o[k])) + b4r[k];
            if (sub == 0) {
                if (cell < cells) {
                    float* ho = dec.head_out + (size_t)(m_base + cell) * 8;
                    const u32x4_t a = {__float_as_uint(o[0]), __float_as_uint(o[1]), __float_as_uint(o[2]), __float_as_uint(o[3])};
                    const u32x4_t z = {__float_as_uint(o[4]), 0u, 0u, 0u};
                    // write-through: read by another workgroup of this launch (sc1 loads) after the ticket below
                    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\tglobal_store_dwordx4 %0, %2, off offset:16 sc1"
                                 : : "v"(ho), "v"(a), "v"(z) : "memory");
                }
                if (cell < MB * 16)
                    s_resp[cell] = cell < cells ? hc_sigmoid(o[0]) * hannr[it] : -3.0f;   // padding: never a candidate
            }
        }
        __syncthreads();
        if (wave == 0) {                          // first maximum of the band -> one 8-B candidate
            float best = -2.0f;
            int bidx = 0x7fffffff;
#pragma unroll
            for (int c = 0; c < MB * 16; c += 64) {
                const int cell = c + lane;
                if (cell < MB * 16) {
                    const float r = s_resp[cell];
                    const int gi = y0 * grid + cell;
                    if (hc_better(r, gi, best, bidx)) { best = r; bidx = gi; }
                }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bidx, off);
                if (hc_better(ob, oi, best, bidx)) { best = ob; bidx = oi; }
            }
            if (lane == 0) {
                float* cand = p.band_best + 2 * ((size_t)b * p.bands + band);
                asm volatile("global_store_dwordx2 %0, %1, off sc1" : : "v"(cand), "v"(make_uint2(__float_as_uint(best), (uint32_t)bidx)) : "memory");
            }
        }
        // ---- hand-off: the last band of the stream to arrive decodes it ------------------------------------
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // every storing wave, before the barrier
        __syncthreads();
        int* s_last = reinterpret_cast<int*>(smem);
        if (tid == 0) {
            const unsigned t = __hip_atomic_fetch_add(p.band_cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = t == (unsigned)(p.bands - 1);
            if (last) __hip_atomic_store(p.band_cnt + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last[0] = last;
        }
        __syncthreads();
        if (!s_last[0]) return;                     // workgroup-uniform
        // the stream's argmax from the bands' candidates (one sc1 load per band), then the window and the box
        if (wave == 0) {
            float best = -2.0f;
            int bidx = 0x7fffffff;
            for (int c = lane; c < p.bands; c += 64) {
                uint2 v;
                asm volatile("global_load_dwordx2 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)"
                             : "=&v"(v) : "v"(p.band_best + 2 * ((size_t)b * p.bands + c)) : "memory");
                const float r = __uint_as_float(v.x);
                if (hc_better(r, (int)v.y, best, bidx)) { best = r; bidx = (int)v.y; }
            }
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float ob = __shfl_xor(best, off);
                const int oi = __shfl_xor(bidx, off);
                if (hc_better(ob, oi, best, bidx)) { best = ob; bidx = oi; }
            }
            if (lane == 0) s_last[1] = bidx;
        }
        __syncthreads();
        const int idx = s_last[1];
        StreamState pre_state;
        PassOut pre_po;
        if (tid == 0) {
            vm_drain();
            asm volatile("s_waitcnt vmcnt(0)" : "+v"(pre_s[0]), "+v"(pre_s[1]), "+v"(pre_s[2]), "+v"(pre_s[3]), "+v"(pre_s[4]),
                         "+v"(pre_s[5]), "+v"(pre_s[6]), "+v"(pre_s[7]), "+v"(pre_s[8]), "+v"(pre_s[9]), "+v"(pre_s[10]),
                         "+v"(pre_o[0]), "+v"(pre_o[1]) : : "memory");
            uint32_t w[22];
#pragma unroll
            for (int i = 0; i < 11; ++i) { w[2 * i] = pre_s[i].x; w[2 * i + 1] = pre_s[i].y; }
            __builtin_memcpy(&pre_state, w, 88);
            const uint32_t q[4] = {pre_o[0].x, pre_o[0].y, pre_o[1].x, pre_o[1].y};
            __builtin_memcpy(&pre_po, q, 16);
        }
        decode_box<true, true>(dec, b, idx, tid, reinterpret_cast<float*>(smem + 64), pre_state, pre_po);
    }
}

// ---- launch plan -----------------------------------------------------------------------------------------
// Rows per band R and column groups: one round of workgroups over the 256 CUs where the batch allows it, and
// the least bytes through one CU (the weights of its column group + its A image): see the header.
// LDS bytes of a launch (0: the band does not fit): [A image | max(ring, staged output tile)]
static size_t headconv_lds(int grid, int C, int N, int K, bool halo, bool tail, bool ln, int R, int ncb) {
    const int BN = 64 * ncb, mb = (R * grid + 15) / 16, mbe = (mb + 1) & ~1;
    const int BK = halo ? C : 64, rowb = BK * 2, rpp = 1024 / rowb;
    const size_t himg = halo ? (size_t)(((R + 2) * (grid + 2) + rpp - 1) / rpp) * rpp * rowb
                             : ln ? (((size_t)(K / 64) * R * grid * rowb) + 1023) & ~(size_t)1023 : 0;
    const size_t stage = ((halo || ln) ? 0 : (size_t)mbe * 16 * rowb) + (size_t)BN * rowb;
    const size_t ring = ((halo || ln) ? 3 : 4) * stage;
    const size_t otile = (size_t)mb * 16 * (BN * 2 + 16);
    size_t smem = himg + std::max(ring, std::max(otile, (size_t)8192));
    // the fused tail's hand-off (write-through stores, ticket, sc1 loads in place of an acquire) is the form the MI355X
    // guide measured with ONE workgroup per CU: small models would fit several, so the tail asks for more than half a
    // CU's LDS (as the 256x256 GEMM's row-panel hand-off has by construction)
    if (tail) smem = std::max(smem, (size_t)84 * 1024);
    return smem <= 160 * 1024 ? smem : 0;
}

static void headconv_plan(int B, int grid, int C, int N, int K, bool halo, bool tail, bool ln, int* R_out, int* ncb_out) {
    double best = 1e30;
    int bestR = 0, bestncb = N >= 128 ? 2 : 1;
    for (int ncb = (N % 128 == 0) ? 2 : 1; ncb >= 1; --ncb) {
        const int BN = 64 * ncb;
        if (N % BN) continue;
        if (tail && BN != N) continue;
        for (int R = 1; R <= grid; ++R) {
            const int mb = (R * grid + 15) / 16;
            if (mb > HC_MBMAX) break;
            if (!headconv_lds(grid, C, N, K, halo, tail, ln, R, ncb)) break;
            const int bands = (grid + R - 1) / R;
            const long wgs = (long)B * bands * (N / BN);
            const long rounds = (wgs + 255) / 256;
            const double wbytes = (double)BN * K * 2;
            const double abytes = halo ? (double)(R + 2) * (grid + 2) * C * 2 : (double)mb * 16 * K * (ln ? 4 : 2);
            const double stream_us = (wbytes + abytes) / 70e3;              // ~70 GB/s per CU from L2
            const double mfma_us = (double)mb * ncb * (K / 32) * 16 / 1.9e3;  // 16 cycles per MFMA, one wave per SIMD
            const double t = rounds * (std::max(stream_us, mfma_us) + 0.5) + 0.002 * bands;
            if (t < best) { best = t; bestR = R; bestncb = ncb; }
        }
    }
    *R_out = bestR; *ncb_out = bestncb;
}

template <int BK, int NCB, int MB, bool HALO, bool TAIL, int LNC>
static hipError_t headconv_launch_t(const HeadConvArgs& a, const DecodeArgs& dec, int wgs, size_t smem, hipStream_t st) {
    vt_launch((head_conv_kernel<BK, NCB, MB, HALO, TAIL, LNC>), dim3(wgs), dim3(512), smem, st, a, dec);
    return hipGetLastError();
}
template <int BK, int NCB, bool HALO, bool TAIL, int LNC = 0>
static hipError_t headconv_launch_mb(int mb, const HeadConvArgs& a, const DecodeArgs& dec, int wgs, size_t smem, hipStream_t st) {
    switch (mb) {
        case 1: return headconv_launch_t<BK, NCB, 1, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        case 2: return headconv_launch_t<BK, NCB, 2, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        case 3: return headconv_launch_t<BK, NCB, 3, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        case 4: return headconv_launch_t<BK, NCB, 4, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        case 5: return headconv_launch_t<BK, NCB, 5, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        case 6: return headconv_launch_t<BK, NCB, 6, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        case 7: return headconv_launch_t<BK, NCB, 7, HALO, TAIL, LNC>(a, dec, wgs, smem, st);
        default: return hipErrorInvalidValue;
    }
}

template <int BK, int NCB, int MB, bool HALO, bool TAIL, int LNC>
static hipError_t headconv_prep_t() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&head_conv_kernel<BK, NCB, MB, HALO, TAIL, LNC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}
template <int BK, int NCB, bool HALO, bool TAIL, int LNC = 0>
static hipError_t headconv_prep_mb() {
    hipError_t e = headconv_prep_t<BK, NCB, 1, HALO, TAIL, LNC>();
    if (e == hipSuccess) e = headconv_prep_t<BK, NCB, 2, HALO, TAIL, LNC>();
    if (e == hipSuccess) e = headconv_prep_t<BK, NCB, 3, HALO, TAIL, LNC>();
    if (e == hipSuccess) e = headconv_prep_t<BK, NCB, 4, HALO, TAIL, LNC>();
    if (e == hipSuccess) e = headconv_prep_t<BK, NCB, 5, HALO, TAIL, LNC>();
    if (e == hipSuccess) e = headconv_prep_t<BK, NCB, 6, HALO, TAIL, LNC>();
    if (e == hipSuccess) e = headconv_prep_t<BK, NCB, 7, HALO, TAIL, LNC>();
    return e;
}

// Raise the dynamic-LDS limit of every instantiation once per device, outside any stream capture.
hipError_t headconv_prepare() {
    hipError_t e = headconv_prep_mb<128, 1, true, false>();
    if (e == hipSuccess) e = headconv_prep_mb<128, 2, true, false>();
    if (e == hipSuccess) e = headconv_prep_mb<128, 2, true, true>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 1, true, false>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 1, true, true>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 1, false, false>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 2, false, false>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 1, false, false, 3>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 2, false, false, 3>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 1, false, false, 4>();
    if (e == hipSuccess) e = headconv_prep_mb<64, 2, false, false, 4>();
    return e;
}

bool headconv_supported(int grid, int C, int N, int K, bool conv3x3) {
    if (grid < 1 || grid > 16 * HC_MBMAX) return false;
    if (conv3x3) return (C == 64 || C == 128) && N == C && K == 9 * C;
    return (N == 64 || N == 128) && K % 64 == 0 && K >= 64;
}

// supported AND plannable: some band height fits LDS for this layer (3x3 at C = 128 and maps of roughly 84..112 cells
// per row: the halo image beside the 96-KiB ring, or the tail's 84-KiB floor, exceeds 160 KiB even at R = 1). The engine
// asks this for all three layer kinds before it routes the head to this kernel; a shape that fails takes the
// implicit-GEMM head.
bool headconv_plannable(int grid, int C, int N, int K, bool conv3x3, bool tail) {
    if (!headconv_supported(grid, C, N, K, conv3x3) || (tail && !conv3x3)) return false;
    int R = 0, ncb = 0;
    headconv_plan(1, grid, C, N, K, conv3x3, tail, false, &R, &ncb);
    return R > 0;
}

// the 1x1 layer with the final LayerNorm inside (a.xh / a.xl / a.ln_g / a.ln_b set): D = 768 or 1024, and a band of at
// least one map row must fit LDS as a resident image
bool headconv_ln_supported(int grid, int N, int D) {
    if (!headconv_supported(grid, N, N, D, false) || (D != 768 && D != 1024)) return false;
    int R = 0, ncb = 0;
    headconv_plan(1, grid, N, N, D, false, false, true, &R, &ncb);
    return R > 0;
}

// a.R <= 0: planned here. dec != nullptr: the fused tail (logits + decode) behind a 3x3 layer.
hipError_t launch_headconv(HeadConvArgs a, const DecodeArgs* dec, hipStream_t st) {
    const bool halo = a.conv3x3 != 0, tail = dec != nullptr, ln = a.xh != nullptr;
    if (a.B < 1 || (!a.in && !ln) || !a.W || !a.bias || !a.out) return hipErrorInvalidValue;
    if (!headconv_supported(a.grid, a.C, a.N, a.K, halo)) return hipErrorInvalidValue;
    if (ln && (halo || !a.xl || !a.ln_g || !a.ln_b || a.in_stride < a.in_off + a.grid * a.grid || a.in_off < 0 ||
               !headconv_ln_supported(a.grid, a.N, a.K)))
        return hipErrorInvalidValue;
    if (halo && (!a.zeros || a.ldin < a.C)) return hipErrorInvalidValue;
    if (tail && (!halo || !a.band_cnt || !a.band_best || !dec->w4 || !dec->b4 || !dec->head_out || !dec->hann || !dec->states ||
                 !dec->results || !dec->out || dec->ns != a.grid * a.grid || dec->B != a.B || dec->grid != a.grid))
        return hipErrorInvalidValue;
    int R = a.R, ncb = a.ncb;
    if (R <= 0 || ncb <= 0) headconv_plan(a.B, a.grid, a.C, a.N, a.K, halo, tail, ln, &R, &ncb);
    if (R <= 0) return hipErrorInvalidValue;
    if (R > a.grid) R = a.grid;
    const int BN = 64 * ncb, mbmax = (R * a.grid + 15) / 16;
    if (mbmax > HC_MBMAX || a.N % BN || (tail && BN != a.N)) return hipErrorInvalidValue;
    a.R = R; a.ncb = ncb;
    a.bands = (a.grid + R - 1) / R;
    a.mbe_max = (mbmax + 1) & ~1;
    const int BK = halo ? a.C : 64;
    const size_t smem = headconv_lds(a.grid, a.C, a.N, a.K, halo, tail, ln, R, ncb);
    if (!smem) return hipErrorInvalidValue;
    const int wgs = a.B * a.bands * (a.N / BN);
    DecodeArgs d{};
    if (dec) d = *dec;
    const int mb = mbmax;
    if (halo && BK == 128) {
        if (tail) return headconv_launch_mb<128, 2, true, true>(mb, a, d, wgs, smem, st);
        return ncb == 2 ? headconv_launch_mb<128, 2, true, false>(mb, a, d, wgs, smem, st)
                        : headconv_launch_mb<128, 1, true, false>(mb, a, d, wgs, smem, st);
    }
    if (halo) {
        return tail ? headconv_launch_mb<64, 1, true, true>(mb, a, d, wgs, smem, st)
                    : headconv_launch_mb<64, 1, true, false>(mb, a, d, wgs, smem, st);
    }
    if (ln && a.K == 768)
        return ncb == 2 ? headconv_launch_mb<64, 2, false, false, 3>(mb, a, d, wgs, smem, st)
                        : headconv_launch_mb<64, 1, false, false, 3>(mb, a, d, wgs, smem, st);
    if (ln)
        return ncb == 2 ? headconv_launch_mb<64, 2, false, false, 4>(mb, a, d, wgs, smem, st)
                        : headconv_launch_mb<64, 1, false, false, 4>(mb, a, d, wgs, smem, st);
    return ncb == 2 ? headconv_launch_mb<64, 2, false, false>(mb, a, d, wgs, smem, st)
                    : headconv_launch_mb<64, 1, false, false>(mb, a, d, wgs, smem, st);
}
