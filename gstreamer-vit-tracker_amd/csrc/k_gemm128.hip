// k_gemm128.hip — 128x192-tile, 4-wave bf16 MFMA GEMM for gfx950, TWO workgroups per CU (config 16).
//
//   C[M,N] = A[M,K] · W[N,K]^T with the fused epilogues of k_gemm.hip; N a multiple of 192.
//
// Why a third kernel. In-kernel stamps of the 256x256 kernels (profiles/README.md, round 3): their main
// loop keeps the matrix pipe 96 % busy, but a workgroup owns the whole CU, so nothing runs under its
// epilogue - 28 % of fc1's cycles (GELU: vector-ALU bound), half of proj's (residual read-modify-write:
// HBM bound, every CU in that phase at the same time). Here a workgroup is HALF a CU: 4 waves, one per
// SIMD, 80 KiB of LDS, <= 256 VGPRs; two of them share a CU and are scheduled independently, so one's
// epilogue (vector ALU / HBM) runs under the other's main loop (matrix pipe), and one's barrier, LDS-read
// and staging latencies are filled by the other's MFMAs.
//
// Geometry
//   * 4 waves side by side along N; a wave owns 128 x 48 of C = 8 x 3 v_mfma_f32_16x16x32_bf16
//     accumulators (96 VGPRs), acc[i][mf][nf]: rows i*64 + mf*16 + (lane & 15), columns
//     wave*48 + nf*16 + 4*(lane >> 4) + e (lane = row, registers = 4 consecutive columns).
//   * K-step 64 (whole 128-B lines per LDS-DMA request: the first version of this kernel, 128x256 tiles with
//     a 32-deep step, was bound by the L2 REQUEST rate - profiles/r03_gemm128_experiment.txt). One stage =
//     A 128 rows x 128 B (16 KiB) + W 192 rows x 128 B (24 KiB); two stages = 80 KiB, i.e. BM + BN <= 320 is
//     what two workgroups per CU allow. LDS image as in k_gemm256.hip: 16-B chunk c of row r stored at
//     c ^ ((r >> 1) & 7), XOR on the per-lane SOURCE address of the LDS-DMA and again on the ds_read_b128.
//   * per K-step: wait vmcnt(0) (this step's ten LDS-DMA instructions, issued one step ago, have landed) |
//     barrier (every wave is past the previous step's MFMAs, which consumed its reads of the other buffer) |
//     stage step+1 into the other buffer | per 32-deep half: 11 ds_read_b128, 24 MFMAs.
//     The staging distance is ONE step: when a wave waits for it, the co-resident workgroup has the pipe.
#include "vt_common.hpp"
#include "k_gemm_util.hpp"

#define G128_BM 128
#define G128_BN 192
#define G128_A_BYTES (G128_BM * 128)
#define G128_STAGE ((G128_BM + G128_BN) * 128)
#define G128_LDS (2 * G128_STAGE)

namespace {

typedef f32x4_t acc128_t[2][4][3];   // [i][mf][nf]

template <bool SWAP>
__device__ __forceinline__ f32x4_t mma16(const bf16x8_t& xa, const bf16x8_t& wb, const f32x4_t& c) {
    if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xa, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, wb, c, 0, 0, 0);
}

#ifdef VT_STAMPS
#define G128_T(v) { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define G128_STAMP_ACC() { st[0] += st_c - st_a; st[1] += st_b - st_c; }
#define G128_STAMP_ARG , unsigned long long (&st)[2]
#else
#define G128_T(v)
#define G128_STAMP_ACC()
#define G128_STAMP_ARG
#endif

// SWAP: lane = row of C (l & 15), registers = 4 consecutive columns; else lane = column, registers = rows
template <bool SWAP>
__device__ __forceinline__ void g128_mainloop(const GemmArgs& p, char* smem, int m0, int n0, acc128_t& acc G128_STAMP_ARG) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // LDS-DMA: one instruction of the workgroup fills 32 rows x 128 B; lane L of wave w writes slot
    // (row w*8 + L/8, chunk L & 7) and therefore fetches source chunk (L & 7) ^ ((row >> 1) & 7)
    const int srow = tid >> 3;
    const int sch = (tid & 7) ^ ((srow >> 1) & 7);
    uint32_t aoff[4], boff[6];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int gm = m0 + r * 32 + srow;
        gm = gm < p.M ? gm : p.M - 1;                 // rows past M read the last row; never stored
        aoff[r] = (uint32_t)(((size_t)gm * p.lda + sch * 8) * 2);      // < 2^32: gemm128_fits
    }
#pragma unroll
    for (int r = 0; r < 6; ++r) boff[r] = (uint32_t)(((size_t)(n0 + r * 32 + srow) * p.ldw + sch * 8) * 2);
    const char* const gA = reinterpret_cast<const char*>(p.A);
    const char* const gW = reinterpret_cast<const char*>(p.W);
    char* const dst0 = smem + wave * 1024;
    auto stage = [&](int kt, int bo) {
        const char* const sa = gA + (size_t)kt * 128;
        const char* const sw_ = gW + (size_t)kt * 128;
#pragma unroll
        for (int r = 0; r < 4; ++r) glds16(sa + aoff[r], dst0 + bo + r * 4096);
#pragma unroll
        for (int r = 0; r < 6; ++r) glds16(sw_ + boff[r], dst0 + bo + G128_A_BYTES + r * 4096);
    };
    // fragment reads: lane (l15, q) -> row l15 of the 16-row block, 16-B chunk q (+ 4 for the second half)
    const int l15 = lane & 15, q = lane >> 4, sw = (lane >> 1) & 7;
    const uint32_t lterm = (uint32_t)(l15 * 128 + ((q ^ sw) << 4));
    const char* const fa[2] = {smem + lterm, smem + (lterm ^ 64)};                       // k halves 0 / 1
    const char* const fb[2] = {smem + G128_A_BYTES + wave * (48 * 128) + lterm,
                               smem + G128_A_BYTES + wave * (48 * 128) + (lterm ^ 64)};
    const int nk = p.K >> 6;
#ifdef VT_STAMPS
    unsigned long long st_a, st_b, st_c;
#endif

    stage(0, 0);
    int bo = 0;
    for (int kt = 0; kt < nk; ++kt) {
        G128_T(st_a)
        wait_vmcnt<0>();
        G128_T(st_c)
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        G128_T(st_b)
        G128_STAMP_ACC()
        if (kt + 1 < nk) stage(kt + 1, bo ^ G128_STAGE);
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t Af[2][4], Bf[3];
#pragma unroll
            for (int nf = 0; nf < 3; ++nf)
                Bf[nf] = *reinterpret_cast<const bf16x8_t*>(fb[kk] + bo + nf * 2048);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    Af[i][mf] = *reinterpret_cast<const bf16x8_t*>(fa[kk] + bo + i * 8192 + mf * 2048);
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                    for (int nf = 0; nf < 3; ++nf) acc[i][mf][nf] = mma16<SWAP>(Af[i][mf], Bf[nf], acc[i][mf][nf]);
            __builtin_amdgcn_s_setprio(0);
        }
        bo ^= G128_STAGE;
    }
}

// tile order: XCD-contiguous (k_gemm.hip), then column groups of `cg` tile columns, row panels fastest
// inside a group - the ~64 tiles an XCD works on at a time share cg W tiles and 64 / cg A panels
__device__ __forceinline__ void g128_tile(int tiles_m, int cg, int& m0, int& n0) {
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int per_group = tiles_m * cg;
    const int g = bid / per_group, r = bid % per_group;
    m0 = (r / cg) * G128_BM;
    n0 = (g * cg + r % cg) * G128_BN;
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm128_kernel(GemmArgs p, int cg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_m = (p.M + G128_BM - 1) / G128_BM;
    int m0, n0;
    g128_tile(tiles_m, cg, m0, n0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wc = wave, l15 = lane & 15, q = lane >> 4;

    acc128_t acc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) acc[i][mf][nf] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    static_assert(EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || EPI == EPI_QKV, "bf16 epilogues");
    float scale = 1.0f;
    if constexpr (EPI == EPI_QKV) scale = (n0 < p.D) ? ATT_Q_SCALE : 1.0f;   // q * log2(e)/sqrt(64)
#ifdef VT_STAMPS   // per wave: [vmcnt waits, barrier waits, main loop total, kernel total] in s_memtime ticks
    unsigned long long st[2] = {0, 0};
    const unsigned long long st_t0 = __builtin_amdgcn_s_memtime();
    g128_mainloop<true>(p, smem, m0, n0, acc, st);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long st_t1 = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#else
    g128_mainloop<true>(p, smem, m0, n0, acc);
#endif
    // whole tile as [128 rows][384 B] bf16; 8-B chunk c8 (0..47) of row r stored at c8 ^ ((r & 7) << 1)
    // (the XOR touches bits 1..3 only: it stays inside the row's 48 chunks).
    // folded LayerNorm (vt_common.hpp): y = a_r * acc + (b_r * colsum[n] + bias[n]); without one
    // a_r = 1, b_r = 0 and fma(1, acc, bias) = acc + bias exactly
    const bool ln = p.rowstat != nullptr;
    f32x4_t bias4[3], cs4[3];
#pragma unroll
    for (int nf = 0; nf < 3; ++nf) {
        bias4[nf] = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wc * 48 + nf * 16 + 4 * q);
        cs4[nf] = ln ? *reinterpret_cast<const f32x4_t*>(p.colsum + n0 + wc * 48 + nf * 16 + 4 * q)
                     : f32x4_t{0.f, 0.f, 0.f, 0.f};
    }
    float2 rs[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int row = m0 + i * 64 + mf * 16 + l15;
            rs[i][mf] = ln ? p.rowstat[row < p.M ? row : p.M - 1] : make_float2(1.0f, 0.0f);
        }
    __syncthreads();                              // every wave has finished reading the ring
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const int row = i * 64 + mf * 16 + l15;
#pragma unroll
            for (int nf = 0; nf < 3; ++nf) {
                const int c8 = wc * 12 + nf * 4 + q;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float x = __builtin_fmaf(rs[i][mf].x, acc[i][mf][nf][e],
                                                   __builtin_fmaf(rs[i][mf].y, cs4[nf][e], bias4[nf][e]));
                    if constexpr (EPI == EPI_GELU_BF16) v[e] = gelu_erf(x);
                    else if constexpr (EPI == EPI_RELU_BF16) v[e] = fmaxf(x, 0.0f);
                    else v[e] = x * scale;
                }
                *reinterpret_cast<uint2*>(smem + row * 384 + ((c8 ^ ((row & 7) << 1)) << 3)) =
                    make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
            }
        }
    __syncthreads();
    // read-out: 32 lanes per row, 24 of them carry one of the row's 16-B chunks (three whole 128-B lines)
    const int c16 = tid & 31;
    if (c16 < 24) {
        bf16_t* const obase = (EPI == EPI_QKV) ? p.qk + (size_t)m0 * (2 * p.D) + n0 + c16 * 8
                                               : p.Cb + (size_t)m0 * p.ldcb + n0 + c16 * 8;
        const size_t ostride = (EPI == EPI_QKV) ? (size_t)(2 * p.D) : (size_t)p.ldcb;
        if (m0 + G128_BM <= p.M) {
#pragma unroll
            for (int hb = 0; hb < 2; ++hb) {
                uint4 v[8];
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = (hb * 8 + it) * 8 + (tid >> 5);
                    v[it] = *reinterpret_cast<const uint4*>(smem + row * 384 + ((c16 ^ (row & 7)) << 4));
                }
#pragma unroll
                for (int it = 0; it < 8; ++it) {
                    const int row = (hb * 8 + it) * 8 + (tid >> 5);
                    *reinterpret_cast<uint4*>(obase + (size_t)row * ostride) = v[it];
                }
            }
        } else {
#pragma unroll 4
            for (int it = 0; it < 16; ++it) {
                const int row = it * 8 + (tid >> 5);
                if (m0 + row < p.M) {
                    const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 384 + ((c16 ^ (row & 7)) << 4));
                    *reinterpret_cast<uint4*>(obase + (size_t)row * ostride) = v;
                }
            }
        }
    }
#ifdef VT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = st[0]; d[1] = st[1]; d[2] = st_t1 - st_t0; d[3] = __builtin_amdgcn_s_memtime() - st_t0;
    }
#endif
}

template <int EPI>
hipError_t prepare_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm128_kernel<EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, G128_LDS);
}

// column-group width: the largest divisor of tiles_n up to 8 (64 concurrent tiles of an XCD as ~8 x 8:
// a W tile is 1.5 A panels)
int column_group(int tiles_n) {
    int best = 1;
    for (int c = 1; c <= tiles_n && c <= 8; ++c)
        if (tiles_n % c == 0) best = c;
    return best;
}

template <int EPI>
hipError_t launch_one(const GemmArgs& a, hipStream_t st) {
    const int tiles_n = a.N / G128_BN, tiles_m = (a.M + G128_BM - 1) / G128_BM;
    hipLaunchKernelGGL(gemm128_kernel<EPI>, dim3(tiles_m * tiles_n), dim3(256), G128_LDS, st, a, column_group(tiles_n));
    return hipGetLastError();
}

}  // namespace

hipError_t gemm128_prepare() {
    hipError_t e;
    if ((e = prepare_one<EPI_GELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_one<EPI_RELU_BF16>()) != hipSuccess) return e;
    return prepare_one<EPI_QKV>();
}

bool gemm128_fits(const GemmArgs& a, int epilogue) {
    if (a.M <= 0 || a.N % G128_BN != 0 || a.K % 64 != 0 || a.K < 64) return false;
    if ((long long)a.M * a.lda * 2 >= VT_GEMM256_MAX_OPERAND_BYTES || (long long)a.N * a.ldw * 2 >= VT_GEMM256_MAX_OPERAND_BYTES)
        return false;
    if ((a.lda & 7) || (a.ldw & 7) || a.conv_grid > 0) return false;
    switch (epilogue) {
        case EPI_GELU_BF16:
        case EPI_RELU_BF16:
            return !((a.ldcb & 7) || !a.Cb || !a.bias || (a.rowstat && !a.colsum));
        default: return false;
    }
}

// hipErrorInvalidValue: the shape does not fit this kernel
hipError_t launch_gemm128(const GemmArgs& a, int epilogue, hipStream_t st) {
    if (!gemm128_fits(a, epilogue)) return hipErrorInvalidValue;
    switch (epilogue) {
        case EPI_GELU_BF16: return launch_one<EPI_GELU_BF16>(a, st);
        case EPI_RELU_BF16: return launch_one<EPI_RELU_BF16>(a, st);
        default: return hipErrorInvalidValue;
    }
}
