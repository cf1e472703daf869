// k_gemm256.hip — 256x256-tile, 8-wave, phase-interleaved bf16 MFMA GEMM for gfx950 (configs 18, 19).
//
//   C[M,N] = A[M,K] · W[N,K]^T (+ the fused epilogues of k_gemm.hip), one 256x256 output tile per
//   workgroup, one workgroup per CU (128 KiB of LDS, 512 threads).
//
// Why a second kernel: the 128x128 / 4-wave kernel of k_gemm.hip is bounded by the rate at which a
// CU gets operand tiles from L2 (profiles/README.md: the block parks on s_waitcnt vmcnt for 43 %
// of its cycles). A 256x256 tile moves half the bytes per FLOP, and the schedule below keeps four
// half-tiles (64 KiB) per CU in flight at all times, issued 5-6 phases before they are needed.
//
// Geometry
//   * 8 waves as 2 (M) x 4 (N); a wave owns a 128 x 64 piece of C = 8 x 4 v_mfma_f32_16x16x32_bf16
//     accumulators (128 VGPRs). It is computed as two 64 x 64 halves (rows i = 0, 1), one per PHASE,
//     32 MFMAs each.
//   * K-tile depth 64. One K-tile in LDS = four HALF-tiles of 16 KiB, each 128 rows x 128 B:
//       A_i = the rows every wave uses for quadrant row i  (rows wr*128 + i*64 + 0..63, wr = 0,1)
//       B_j = the W rows every wave uses for quadrant column j (wc*64 + j*32 + 0..31, wc = 0..3)
//     so that a half-tile is dead as soon as its phase has read it (B0 and B1 stay in registers over
//     both phases). Two K-tile buffers (128 KiB).
//   * LDS image as in k_gemm.hip: 128-B rows, 16-B chunk c of row r stored at c ^ ((r >> 1) & 7);
//     LDS-DMA (global_load_lds_dwordx4) writes lane-linear, so the XOR sits on the per-lane SOURCE
//     address and again on the ds_read_b128 (conflict-free for 16-row fragments too: the 16 lanes
//     of each read group land on 16 distinct 16-B slots of the 256-B bank row).
//
// Schedule: two phases of 32 MFMAs per K-tile, documented at g256_mainloop2 below (RAW / WAR argument there).
//   The two wave rows run one barrier apart (wr = 1 executes one extra s_barrier before the loop,
//   wr = 0 one after it), so on every SIMD one wave is in its MFMA section while the other reads
//   LDS and issues loads.
//
// Epilogues go through LDS (dead after the loop) so that global stores are whole rows:
// f32 outputs in two passes of 128 rows x 1 KiB, bf16 outputs in one pass of 256 rows x 512 B.
#include <type_traits>
#include "vt_common.hpp"
#include "k_gemm_util.hpp"

#define G256_BUF 65536
#define G256_HALF 16384
#define G256_LDS (2 * G256_BUF)

namespace {

typedef f32x4_t acc256_t[2][4][2][2];   // [i][mf][j][nf]

template <bool SWAP>
__device__ __forceinline__ f32x4_t mma16(const bf16x8_t& xa, const bf16x8_t& wb, const f32x4_t& c) {
    // SWAP: D = W-frag · X-frag^T, lane = row of C (l & 15), registers = 4 consecutive columns
    // else: D = X-frag · W-frag^T, lane = column of C,       registers = 4 consecutive rows
    if constexpr (SWAP) return __builtin_amdgcn_mfma_f32_16x16x32_bf16(wb, xa, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa, wb, c, 0, 0, 0);
}

// (the A operand's LDS-DMA with the non-temporal hint was measured in round 4 and lost: A then loses the in-step reuse
// by the six workgroups of a panel - profiles/r04_fc1_tile_order_cache_policy.txt; git history has the variant)
#define G256_GLDS_A glds16
#define G256_RD(ptr, off) (*reinterpret_cast<const bf16x8_t*>((ptr) + (off)))

// LDS accesses of the persistent kernel's wave-private epilogue, as inline asm: hipcc (ROCm 7.2) puts
// an s_waitcnt vmcnt(0) in front of compiler-visible LDS reads that follow the epilogue's LDS-DMA
// prologue pieces and global stores, which drains both in every block. Operations without a VGPR
// destination are register-safe; the reads are form (ii) of the guide (s5.7): "=v" destinations, and
// one wait statement naming every destination "+v" before the first consumer.
__device__ __forceinline__ uint32_t lds_addr(const void* p) {
    return (uint32_t)(uintptr_t)((const __attribute__((address_space(3))) char*)p);
}
template <int OFF>
__device__ __forceinline__ void lds_write64_asm(uint32_t addr, uint32_t lo, uint32_t hi) {
    const unsigned long long v = ((unsigned long long)hi << 32) | lo;
    asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read128_asm(u32x4_t& dst, uint32_t addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read64_asm(unsigned long long& dst, uint32_t addr) {
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait2_asm(u32x4_t& a, u32x4_t& b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)::"memory");
}

#define G256_READ_A(I)                                                        \
    _Pragma("unroll") for (int mf = 0; mf < 4; ++mf) {                        \
        Af[mf][0] = G256_RD(pa0, (I) * G256_HALF + mf * 2048);                \
        Af[mf][1] = G256_RD(pa1, (I) * G256_HALF + mf * 2048);                \
    }
#define G256_READ_B(J)                                                        \
    _Pragma("unroll") for (int nf = 0; nf < 2; ++nf) {                        \
        Bf[J][nf][0] = G256_RD(pb0, (J) * G256_HALF + nf * 2048);             \
        Bf[J][nf][1] = G256_RD(pb1, (J) * G256_HALF + nf * 2048);             \
    }
// half-tile h of K-tile KT into the buffer at byte offset BO: 2 LDS-DMA instructions per thread
#define G256_STAGE_A(I, KT, BO)                                                                  \
    {                                                                                            \
        const char* s_ = reinterpret_cast<const char*>(p.A) + (size_t)(KT) * 128;                \
        char* d_ = smem + (BO) + (I) * G256_HALF + wave * 1024;                                  \
        G256_GLDS_A(s_ + ((I) ? aoff10 : aoff00), d_);                                           \
        G256_GLDS_A(s_ + ((I) ? aoff11 : aoff01), d_ + 8192);                                    \
    }
#define G256_STAGE_B(J, KT, BO)                                                                  \
    {                                                                                            \
        const char* s_ = reinterpret_cast<const char*>(p.W) + (size_t)(KT) * 128;                \
        char* d_ = smem + (BO) + 2 * G256_HALF + (J) * G256_HALF + wave * 1024;                  \
        glds16(s_ + ((J) ? boff10 : boff00), d_);                                                \
        glds16(s_ + ((J) ? boff11 : boff01), d_ + 8192);                                         \
    }
// ---- the schedule: two LONG phases per K-tile (32 MFMAs each) -----------------------------------------
// Round 1's schedule had four phases of 16 MFMAs per K-tile (git history: config 17). In-kernel stamps of it
// (profiles/README.md): each of its 8 barrier intervals per K-tile lasted max(load section 232, MFMA section 289)
// + ~97 cycles of barrier latency for 256 cycles of matrix-pipe work (66 %). Halving the number of intervals
// halves that fixed cost:
//     P1(t): read B0, B1, A0(t)   stage A0, A1(t+1) -> other buffer   wait vmcnt(8): A1(t) landed
//            lgkmcnt(0) | barrier | 32 MFMA: rows i = 0, both column halves | barrier
//     P2(t): read A1(t)           stage B0, B1(t+2) -> this buffer    wait vmcnt(6): B(t+1), A0(t+1)
//            lgkmcnt(0) | barrier | 32 MFMA: rows i = 1                  | barrier
//   (same two wave rows one barrier apart; same LDS image, staging pieces and fragment addresses).
//   RAW: B(t+1) and A0(t+1) are read in P1(t+1), retired by the vmcnt(6) of P2(t) (the phase before);
//        A1(t) is read in P2(t), retired by the vmcnt(8) of P1(t). LDS-DMA retires in issue order:
//        at P1(t)'s wait the queue is [A1(t) x2 | B(t+1) x4 | A(t+1) x4], at P2(t)'s
//        [B(t+1) x4, A0(t+1) x2 | A1(t+1) x2, B(t+2) x4].
//   WAR: a half-tile is restaged ONE phase after its last read (B(t+2) in P2(t) over B(t), read in
//        P1(t)); that is safe only because every wave drains its LDS reads (lgkmcnt(0)) BEFORE the
//        first barrier of the reading phase, and the other wave row is exactly one barrier behind.
#define G256_COMPUTE2(I)                                                                         \
    {                                                                                            \
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_barrier();                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_setprio(1);                                                           \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                         \
            _Pragma("unroll") for (int mf = 0; mf < 4; ++mf)                                     \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                    \
                    _Pragma("unroll") for (int nf = 0; nf < 2; ++nf)                             \
                        acc[I][mf][j][nf] = mma16<SWAP>(Af[mf][kk], Bf[j][nf][kk], acc[I][mf][j][nf]); \
        __builtin_amdgcn_s_setprio(0);                                                           \
        __builtin_amdgcn_sched_barrier(0);                                                       \
        __builtin_amdgcn_s_barrier();                                                            \
        __builtin_amdgcn_sched_barrier(0);                                                       \
    }

template <bool SWAP>
__device__ __forceinline__ void g256_mainloop2(const GemmArgs& p, char* smem, int m0, int n0,
                                               acc256_t& acc) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3;
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    auto a_off = [&](int i, int g) -> uint32_t {
        int gm = m0 + g * 128 + i * 64 + srow;
        gm = gm < p.M ? gm : p.M - 1;   // rows past M read the last row; never stored
        return (uint32_t)(((size_t)gm * p.lda + schunk * 8) * 2);      // < 2^32: launch_gemm256
    };
    auto b_off = [&](int j, int g) -> uint32_t {
        const int gn = n0 + (g * 2 + (srow >> 5)) * 64 + j * 32 + (srow & 31);
        return (uint32_t)(((size_t)gn * p.ldw + schunk * 8) * 2);
    };
    const uint32_t aoff00 = a_off(0, 0), aoff01 = a_off(0, 1), aoff10 = a_off(1, 0), aoff11 = a_off(1, 1);
    const uint32_t boff00 = b_off(0, 0), boff01 = b_off(0, 1), boff10 = b_off(1, 0), boff11 = b_off(1, 1);
    const int l15 = lane & 15, q = lane >> 4, sw = (lane >> 1) & 7;
    const uint32_t a_k0 = (uint32_t)((wr * 64 + l15) * 128 + ((q ^ sw) << 4));
    const uint32_t b_k0 = (uint32_t)(2 * G256_HALF + (wc * 32 + l15) * 128 + ((q ^ sw) << 4));
    const char* pa0 = smem + a_k0;
    const char* pa1 = smem + (a_k0 ^ 64);
    const char* pb0 = smem + b_k0;
    const char* pb1 = smem + (b_k0 ^ 64);

    bf16x8_t Af[4][2], Bf[2][2][2];
    const int nk = p.K >> 6;

    // prologue in the queue order of the steady state: B(0), A(0), then B(1)
    G256_STAGE_B(0, 0, 0)
    G256_STAGE_B(1, 0, 0)
    G256_STAGE_A(0, 0, 0)
    G256_STAGE_A(1, 0, 0)
    G256_STAGE_B(0, 1, G256_BUF)
    G256_STAGE_B(1, 1, G256_BUF)
    wait_vmcnt<6>();                       // B(0), A0(0) landed; A1(0) and B(1) may still fly
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier();   // second wave row runs one barrier behind
    __builtin_amdgcn_sched_barrier(0);

    int bo = 0;
    int kt = 0;
#define G256_FLIP()                                                           \
    bo ^= G256_BUF;                                                           \
    pa0 = smem + (a_k0 ^ bo); pa1 = smem + ((a_k0 ^ 64) ^ bo);                \
    pb0 = smem + (b_k0 ^ bo); pb1 = smem + ((b_k0 ^ 64) ^ bo);
    for (; kt < nk - 2; ++kt) {
        const int bn = bo ^ G256_BUF;
        G256_READ_B(0) G256_READ_B(1) G256_READ_A(0)
        G256_STAGE_A(0, kt + 1, bn)
        G256_STAGE_A(1, kt + 1, bn)
        wait_vmcnt<8>();
        G256_COMPUTE2(0)
        G256_READ_A(1)
        G256_STAGE_B(0, kt + 2, bo)
        G256_STAGE_B(1, kt + 2, bo)
        wait_vmcnt<6>();
        G256_COMPUTE2(1)
        G256_FLIP()
    }
    {   // tile nk-2: B(nk) does not exist
        const int bn = bo ^ G256_BUF;
        G256_READ_B(0) G256_READ_B(1) G256_READ_A(0)
        G256_STAGE_A(0, kt + 1, bn)
        G256_STAGE_A(1, kt + 1, bn)
        wait_vmcnt<8>();
        G256_COMPUTE2(0)
        G256_READ_A(1)
        wait_vmcnt<2>();                   // B(nk-1), A0(nk-1); A1(nk-1) may still fly
        G256_COMPUTE2(1)
        G256_FLIP()
    }
    {   // last tile
        G256_READ_B(0) G256_READ_B(1) G256_READ_A(0)
        wait_vmcnt<0>();                   // A1
        G256_COMPUTE2(0)
        G256_READ_A(1)
        G256_COMPUTE2(1)
    }
#undef G256_FLIP
    if (wr == 0) __builtin_amdgcn_s_barrier();   // pairs with the extra barrier of wave row 1
    __builtin_amdgcn_sched_barrier(0);
}

template <int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tiles_n = p.N >> 8;
    int bid = blockIdx.x;
    {   // XCD-contiguous tile order (see k_gemm.hip); bijective for any grid size
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    const int m0 = (bid / tiles_n) << 8;
    const int n0 = (bid % tiles_n) << 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, q = lane >> 4;

    acc256_t acc;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) acc[i][mf][j][nf] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if constexpr (EPI == EPI_F32_POS || EPI == EPI_RESID || EPI == EPI_F32) {
        // the bias is the accumulators' initial value (lane = row, registers = 4 consecutive columns): one
        // add per element less in an epilogue whose vector work is not hidden by anything
        if (p.bias) {
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    const f32x4_t b4 = *reinterpret_cast<const f32x4_t*>(p.bias + n0 + wc * 64 + j * 32 + nf * 16 + 4 * q);
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf) acc[i][mf][j][nf] = b4;
                }
        }
#ifdef VT_STAMPS
        const unsigned long long xs_t0 = __builtin_amdgcn_s_memtime(), xs_r0 = __builtin_amdgcn_s_memrealtime();
        (void)xs_r0;
#endif
        g256_mainloop2<true>(p, smem, m0, n0, acc);
#ifdef VT_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long xs_t1 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- X-epilogues: the residual stream as the 3-byte pair (bf16 + lo8) + chunk statistics (vt_common.hpp) ----
        // Two passes (i = 0, 1) of 128 rows x 256 f32 staged in LDS: row lr = wr*64 + mf*16 + l15 of the
        // pass, 16-B chunk ch of the row stored at ch ^ (lr & 7). Written out row-wise, 8 consecutive
        // columns per lane (two rows per wave instruction: 512 B of hi and 512 B of lo per row): addend,
        // bias, statistics of the 32-column chunk a quad of lanes covers, split, two 16-B stores.
        // The residual read-modify-write is HBM traffic (128 KB in + 128 KB out per tile) on top of the
        // main loop: the addend loads of a whole pass (8 row pairs per wave, 64 VGPRs - the operand
        // fragments are dead) are issued BEFORE the accumulators are staged, and those of pass 1 before
        // pass 0 is written out, so their latency runs under LDS staging and the stores.
        const int c8 = lane & 31, rsub = lane >> 5, n8 = n0 + c8 * 8;
        const int nchunk = p.N / VT_STAT_CHUNK;
        // row pair `it` (0..7) of pass i that this wave writes out: pass rows lr = it*16 + wave*2 + rsub
        auto out_row = [&](int i, int it) { const int lr = it * 16 + wave * 2 + rsub; return m0 + (lr >> 6) * 128 + i * 64 + (lr & 63); };
        // Round 4: the epilogue issues ~2,700 instructions per wave and tile, two waves per SIMD, with the matrix
        // pipe idle - its vector issue is half of its time. The pair and the chunk partials are addressed as a
        // UNIFORM row-pair base (scalar ALU; the wave index through readfirstlane) + a 32-bit per-lane offset,
        // which hipcc turns into the saddr form of global_load / global_store: no 64-bit vector address
        // arithmetic. Whether a row pair lies inside M is a scalar test (both rows valid: plain stores, no
        // exec-mask branch; M is even for every engine, so only row pairs entirely past M take the guarded path).
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);
        const uint32_t lane_off = (uint32_t)(rsub * p.ldx + n8) * 2u;                       // bytes inside the row pair
        const uint32_t lane_off0 = (uint32_t)n8 * 2u;                                       // ... with the row clamped to the pair's first
        const uint32_t lane_off_c = (uint32_t)(rsub * nchunk + (n8 / VT_STAT_CHUNK)) * 8u;   // float2 entries
        auto row_u = [&](int i, int it) { return m0 + (it >> 2) * 128 + i * 64 + (it & 3) * 16 + wave_u * 2; };   // uniform
        auto load_addend = [&](int i, int it, u32x4_t& a0, u32x4_t& a1) {
            if constexpr (EPI == EPI_RESID) {
                // rows past M read a valid row (value unused): the pair's base clamped on the scalar side, the
                // second row folded onto the first where it alone is past M
                const int mu = row_u(i, it), mc = mu < p.M ? mu : p.M - 1;
                const size_t ro = (size_t)mc * p.ldx * 2;
                const uint32_t lo_ = (mc + 1 < p.M) ? lane_off : lane_off0;
                a0 = *reinterpret_cast<const u32x4_t*>(reinterpret_cast<const char*>(p.Xh) + ro + lo_);
                // the lo8 plane: one byte per element, the same (row, column) at half the byte offset: 8 B per lane
                const u32x2_t l8 = *reinterpret_cast<const u32x2_t*>(reinterpret_cast<const char*>(p.Xl) + (ro >> 1) + (lo_ >> 1));
                a1 = u32x4_t{l8[0], l8[1], 0u, 0u};
            } else if constexpr (EPI == EPI_F32_POS) {
                const int m = out_row(i, it);
                const int mc = m < p.M ? m : p.M - 1;     // clamped: every address valid, value unused
                const float* src = p.pos + (size_t)(mc % p.pos_rows) * p.ldx + n8;
                a0 = *reinterpret_cast<const u32x4_t*>(src);
                a1 = *reinterpret_cast<const u32x4_t*>(src + 4);
            } else {
                a0 = u32x4_t{0u, 0u, 0u, 0u}; a1 = a0;
            }
        };
        auto stage_pass = [&](int i) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) {
                const int lr = wr * 64 + mf * 16 + l15;
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf) {
                        const int ch = wc * 16 + j * 8 + nf * 4 + q;
                        *reinterpret_cast<f32x4_t*>(smem + lr * 1024 + ((ch ^ (lr & 7)) << 4)) =
                            acc[i][mf][j][nf];
                    }
            }
        };
        auto write_pass = [&](int i, const u32x4_t (&ad)[8][2]) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int lr = it * 16 + wave * 2 + rsub;
                const f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(smem + lr * 1024 + (((2 * c8) ^ (lr & 7)) << 4));
                const f32x4_t f1 = *reinterpret_cast<const f32x4_t*>(smem + lr * 1024 + (((2 * c8 + 1) ^ (lr & 7)) << 4));
                float add[8], x[8];
                if constexpr (EPI == EPI_RESID) {
                    x_join8(ad[it][0], u32x2_t{ad[it][1][0], ad[it][1][1]}, add);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { add[e] = __uint_as_float(ad[it][0][e]); add[4 + e] = __uint_as_float(ad[it][1][e]); }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    x[e] = f0[e] + add[e];
                    x[4 + e] = f1[e] + add[4 + e];
                }
                float csum, cm2;
                x_chunk_stats(x, csum, cm2);
                u32x4_t hi;
                u32x2_t lo;
                x_split8(x, hi, lo);
                // chunk partials: two chunks (this quad's and the next one's) per 16-B WRITE-THROUGH (sc1) store -
                // the row panel's last workgroup may read them in this launch (finalize below); as 8-B sc1 stores
                // they were 518 k single fabric writes per launch (an 8-B write-through store costs 2.7x a 16-B
                // one per byte): + 5 us on fc2. The next quad's partials (lane + 4, same 16-lane row: the storing
                // lanes sit at 0 and 8 of a row) come by DPP row_shl:4 - one vector move each instead of a
                // ds_bpermute round trip through the LDS.
                const float nsum = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, csum), 0x104, 0xF, 0xF, true));
                const float nm2 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, cm2), 0x104, 0xF, 0xF, true));
                const f32x4_t v4 = {csum, cm2, nsum, nm2};
                const int mu = row_u(i, it);
                const size_t ro = (size_t)mu * p.ldx * 2;
                char* const dh = reinterpret_cast<char*>(p.Xh) + ro + lane_off;
                char* const dl = reinterpret_cast<char*>(p.Xl) + (ro >> 1) + (lane_off >> 1);      // lo8 plane: 8 B per lane
                char* const dc = reinterpret_cast<char*>(p.cstat) + (size_t)mu * nchunk * 8 + lane_off_c;
                const bool cst = p.cstat && (c8 & 7) == 0;
                if (mu + 1 < p.M) {                 // scalar: both rows of the pair inside M
                    *reinterpret_cast<u32x4_t*>(dh) = hi;
                    *reinterpret_cast<u32x2_t*>(dl) = lo;
                    if (cst) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dc), "v"(v4) : "memory");
                } else if (mu + rsub < p.M) {       // the last row of an odd M
                    *reinterpret_cast<u32x4_t*>(dh) = hi;
                    *reinterpret_cast<u32x2_t*>(dl) = lo;
                    if (cst) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dc), "v"(v4) : "memory");
                }
            }
        };
        u32x4_t ad0[8][2], ad1[8][2];
#pragma unroll
        for (int it = 0; it < 8; ++it) load_addend(0, it, ad0[it][0], ad0[it][1]);
        __syncthreads();
        stage_pass(0);
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 8; ++it) load_addend(1, it, ad1[it][0], ad1[it][1]);
        write_pass(0, ad0);
        __syncthreads();
        stage_pass(1);
        __syncthreads();
        write_pass(1, ad1);
#ifdef VT_STAMPS
        __builtin_amdgcn_sched_barrier(0);
        const unsigned long long xs_t2 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        // ---- row terms of the LayerNorm that follows, by the LAST workgroup of the row panel ------------
        // (instead of a launch of their own: 24 per pass, 6.5 us each between 40-90 us GEMMs.) The N / 256
        // workgroups of a 256-row panel each stored their chunk partials write-through; every wave drains
        // its stores, the workgroup's barrier, then ONE lane takes a ticket on the panel's counter
        // (agent-scope add). The workgroup whose add came last - told by the value the add returned - reads
        // the panel's partials with sc1 loads behind a workgroup barrier that lane has joined, combines them
        // per row (parallel-variance form, as launch_rowstat_finalize does) and stores (rstd, -mean * rstd)
        // for the next kernel. That is the first row of the table of hand-offs measured with sc1 loads in
        // place of an acquire (MI355X guide, "Valid forms": one signalling lane per workgroup after every
        // storing wave's vmcnt(0); one 128-KiB-LDS workgroup per CU; 16-B sc1 stores and 16-B sc1 loads, both in that row); no dispatch
        // order, timing or placement is assumed. The counter is left at zero for the next launch.
        if (p.rowstat_out) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            int* s_last = reinterpret_cast<int*>(smem);
            const int tiles_n_ = p.N >> 8, panel = m0 >> 8;
            if (tid == 0) {
                const unsigned t = __hip_atomic_fetch_add(p.panel_cnt + panel, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int last = t == (unsigned)(tiles_n_ - 1);
                if (last) __hip_atomic_store(p.panel_cnt + panel, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                *s_last = last;
            }
            __syncthreads();
            if (*s_last) {
                // two lanes per row, each half of the row's chunks: nchunk / 4 sc1 loads of 16 B (two
                // chunks each), all issued before the one wait (agent-scope atomic loads are waited for
                // one by one by hipcc: 24 dependent round trips at the end of every launch)
                const int row = tid >> 1, hpart = tid & 1, m = m0 + row;
                const int mc = m < p.M ? m : p.M - 1;
                const char* src = reinterpret_cast<const char*>(p.cstat + (size_t)mc * nchunk) + hpart * (nchunk * 4);
                const int nl = nchunk >> 2;                  // N % 256 == 0: nchunk is a multiple of 8
                u32x4_t v[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) {
                    v[i] = u32x4_t{0u, 0u, 0u, 0u};
                    if (i < nl) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v[i]) : "v"(src + i * 16) : "memory");
                }
                asm volatile("s_waitcnt vmcnt(0)"
                             : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]),
                               "+v"(v[8]), "+v"(v[9]), "+v"(v[10]), "+v"(v[11])
                             :
                             : "memory");
                float s = 0.0f;
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    if (i < nl) s += __uint_as_float(v[i][0]) + __uint_as_float(v[i][2]);
                s += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, s), 0xB1, 0xF, 0xF, true));
                const float Dn = (float)p.N, mean = s / Dn;
                float m2 = 0.0f;
#pragma unroll
                for (int i = 0; i < 12; ++i)
                    if (i < nl) {
                        const float d0 = __uint_as_float(v[i][0]) * (1.0f / VT_STAT_CHUNK) - mean;
                        const float d1 = __uint_as_float(v[i][2]) * (1.0f / VT_STAT_CHUNK) - mean;
                        m2 += (__uint_as_float(v[i][1]) + (float)VT_STAT_CHUNK * (d0 * d0)) +
                              (__uint_as_float(v[i][3]) + (float)VT_STAT_CHUNK * (d1 * d1));
                    }
                m2 += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, m2), 0xB1, 0xF, 0xF, true));
                const float rstd = 1.0f / sqrtf(m2 / Dn + p.ln_eps);
                if (hpart == 0 && m < p.M) p.rowstat_out[m] = make_float2(rstd, -mean * rstd);
            }
        }
#ifdef VT_STAMPS     // phases of an X-epilogue launch, per wave: [main loop, epilogue, stats hand-off, total]
        if (p.dbg && lane == 0) {
            const unsigned long long xs_t3 = __builtin_amdgcn_s_memtime();
            unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
            d[0] = xs_t1 - xs_t0; d[1] = xs_t2 - xs_t1; d[2] = xs_t3 - xs_t2; d[3] = xs_t3 - xs_t0;
#ifdef VT_STAMPS_CLOCK   // shader clock during the launch: s_memtime ticks per 100-MHz s_memrealtime tick
            d[0] = __builtin_amdgcn_s_memrealtime() - xs_r0;
#endif
        }
#endif
    } else {
        bool v_tile = false;
        float scale = 1.0f;
        if constexpr (EPI == EPI_QKV) {
            v_tile = n0 >= 2 * p.D;
            scale = (n0 < p.D) ? ATT_Q_SCALE : 1.0f;   // q * log2(e)/sqrt(64): scores in log2 units
        }
        if (!v_tile) {
            g256_mainloop2<true>(p, smem, m0, n0, acc);
            // whole tile as [256 rows][512 B]; 8-B chunk c8 of row r stored at c8 ^ ((r & 7) << 1)
            // folded LayerNorm (vt_common.hpp): y = a_r * acc + (b_r * colsum[n] + bias[n]); without one
            // a_r = 1, b_r = 0 and fma(1, acc, bias) = acc + bias exactly
            const bool ln = p.rowstat != nullptr;
            f32x4_t bias4[2][2], cs4[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    bias4[j][nf] = *reinterpret_cast<const f32x4_t*>(
                        p.bias + n0 + wc * 64 + j * 32 + nf * 16 + 4 * q);
                    cs4[j][nf] = ln ? *reinterpret_cast<const f32x4_t*>(p.colsum + n0 + wc * 64 + j * 32 + nf * 16 + 4 * q)
                                    : f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    const int row = wr * 128 + i * 64 + mf * 16 + l15;
                    float2 rs = make_float2(1.0f, 0.0f);
                    if (ln) rs = p.rowstat[m0 + row < p.M ? m0 + row : p.M - 1];
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            const int c8 = wc * 16 + j * 8 + nf * 4 + q;
                            float v[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float x = __builtin_fmaf(rs.x, acc[i][mf][j][nf][e],
                                                               __builtin_fmaf(rs.y, cs4[j][nf][e], bias4[j][nf][e]));
                                if constexpr (EPI == EPI_GELU_BF16) v[e] = gelu_erf(x);
                                else if constexpr (EPI == EPI_RELU_BF16) v[e] = fmaxf(x, 0.0f);
                                else v[e] = x * scale;
                            }
                            *reinterpret_cast<uint2*>(smem + row * 512 + ((c8 ^ ((row & 7) << 1)) << 3)) =
                                make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                        }
                }
            __syncthreads();
            const int c16 = tid & 31;
            bf16_t* const obase = (EPI == EPI_QKV) ? p.qk + (size_t)m0 * (2 * p.D) + n0 + c16 * 8
                                                   : p.Cb + (size_t)m0 * p.ldcb + n0 + c16 * 8;
            const size_t ostride = (EPI == EPI_QKV) ? (size_t)(2 * p.D) : (size_t)p.ldcb;
            if (m0 + 256 <= p.M) {
                // full tile (all but the last row panel): no per-row branch, so the LDS reads of a
                // batch are all issued before the first store waits for one (the guarded loop below
                // serialises ds_read -> wait -> store per row)
#pragma unroll
                for (int hb = 0; hb < 2; ++hb) {
                    uint4 v[8];
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = (hb * 8 + it) * 16 + (tid >> 5);
                        v[it] = *reinterpret_cast<const uint4*>(smem + row * 512 + ((c16 ^ (row & 7)) << 4));
                    }
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = (hb * 8 + it) * 16 + (tid >> 5);
                        *reinterpret_cast<uint4*>(obase + (size_t)row * ostride) = v[it];
                    }
                }
            } else {
#pragma unroll 4
                for (int it = 0; it < 16; ++it) {
                    const int row = it * 16 + (tid >> 5);
                    if (m0 + row < p.M) {
                        const uint4 v = *reinterpret_cast<const uint4*>(smem + row * 512 + ((c16 ^ (row & 7)) << 4));
                        *reinterpret_cast<uint4*>(obase + (size_t)row * ostride) = v;
                    }
                }
            }
        } else if constexpr (EPI == EPI_QKV) {
            // V, transposed per head: Vt[b][h][d][t], t contiguous. Lane = d, registers = 4 tokens.
            g256_mainloop2<false>(p, smem, m0, n0, acc);
            const int heads = p.D >> 6;
            const bool ln = p.rowstat != nullptr;
            float bias_[2][2], cs_[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    const int drow = wc * 64 + j * 32 + nf * 16 + l15;
                    bias_[j][nf] = p.bias[n0 + drow];
                    cs_[j][nf] = ln ? p.colsum[n0 + drow] : 0.0f;
                }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf) {
                    // folded LayerNorm: row terms of tokens 4q .. 4q+3 of this 16-row block, fetched once
                    // per block (not per column group); rows past M read the last row, never stored
                    float2 rs[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int mrow = m0 + wr * 128 + i * 64 + mf * 16 + 4 * q + e;
                        rs[e] = ln ? p.rowstat[mrow < p.M ? mrow : p.M - 1] : make_float2(1.0f, 0.0f);
                    }
                    // 4-token run q of the 16-token group: runs 1 and 2 swap places in the permuted
                    // layout (attn_perm16) (only when streams start on 16-token boundaries; otherwise
                    // the runs are placed one by one in the read-out below)
                    const int qp = (p.vt_perm && !(p.tokens & 15)) ? ((q & 1) << 1 | (q >> 1)) : q;
                    const int c8 = wr * 32 + i * 16 + mf * 4 + qp;
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int nf = 0; nf < 2; ++nf) {
                            const int drow = wc * 64 + j * 32 + nf * 16 + l15;
                            const f32x4_t a = acc[i][mf][j][nf];
                            float y[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                y[e] = __builtin_fmaf(rs[e].x, a[e], __builtin_fmaf(rs[e].y, cs_[j][nf], bias_[j][nf]));
                            *reinterpret_cast<uint2*>(smem + drow * 512 + ((c8 ^ ((drow & 7) << 1)) << 3)) =
                                make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
                        }
                }
            __syncthreads();
            if (p.vt_perm && (p.tokens & 15)) {
                // permuted layout with streams that do not start on 16-token boundaries (e.g. 980
                // tokens): every 4-token run goes to attn_perm16 of ITS stream's token index - 8-B
                // stores, twice as many as the aligned case
                const int c8r = tid & 63;
                for (int it = 0; it < 32; ++it) {
                    const int r = it * 8 + (tid >> 6);
                    const int m = m0 + c8r * 4, nv = n0 + r - 2 * p.D;
                    if (m >= p.M) continue;
                    const int b = m / p.tokens, t = m % p.tokens;        // tokens % 4 == 0: runs never straddle
                    bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + attn_perm16(t);
                    *reinterpret_cast<uint2*>(dst) =
                        *reinterpret_cast<const uint2*>(smem + r * 512 + ((c8r ^ ((r & 7) << 1)) << 3));
                }
                return;
            }
            const int c16 = tid & 31;
            for (int it = 0; it < 16; ++it) {
                const int r = it * 16 + (tid >> 5);
                const int m = m0 + c16 * 8, nv = n0 + r - 2 * p.D;
                if (m >= p.M) continue;
                const int b = m / p.tokens, t = m % p.tokens;
                bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + t;
                const char* src = smem + r * 512 + ((c16 ^ (r & 7)) << 4);
                if (t + 8 <= p.tokens && m + 8 <= p.M && ((t & 7) == 0)) {
                    *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
                } else {
                    // the piece straddles a stream boundary (tokens % 8 != 0) or the end of M
                    for (int e = 0; e < 8 && m + e < p.M; ++e) {
                        const int me = m + e, be = me / p.tokens, te = me % p.tokens;
                        p.vt[((size_t)(be * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + te] =
                            *reinterpret_cast<const bf16_t*>(src + 2 * e);
                    }
                }
            }
        }
    }
}

// ---- persistent variant (config 19): bf16-output GEMMs that need several rounds of the chip ----------
// Measured on fc1 (M = 21,600, N = 3072, K = 768; 1020 tiles = 4 rounds of 256 workgroups): a round
// costs 29 us of which the main loop is 16-17; the rest is per-tile overhead that nothing overlaps
// when a fresh workgroup owns the CU for exactly one tile: ~3 us until the LDS-DMA pipeline is full,
// ~3.6 us of bias/activation/pack + LDS transpose, ~3.7 us of stores that all 256 CUs issue in the same
// microsecond (33 MB burst). Here 256 workgroups stay resident and walk the tiles:
//   * the 12 LDS-DMA pieces of the NEXT tile's prologue are issued right after the last barrier of a
//     tile's main loop, BEFORE its epilogue: the pipeline fill runs under the epilogue;
//   * the epilogue is wave-private: each wave transposes its own 128 x 64 piece through its own 4 KiB
//     of the 32 KiB of LDS the operand ring leaves free (four passes of 32 rows x 128 B, no workgroup
//     barrier), so the ring can be refilled meanwhile and fast waves are not held back;
//   * the CUs drift apart after the first tile, so the stores no longer hit HBM as one burst;
//   * tiles are dealt so that the 32 workgroups sharing an XCD (blockIdx % 8, placement is a speed
//     assumption only) run a compact block of ~(32 / cgw) x cgw tiles per round: fewer distinct A/W
//     panels per XCD than 32 consecutive row-major tiles (11.3 instead of 14.7 for N = 3072).
// QKV column tiles that hold V (transposed scatter) keep the block-wide epilogue through the ring and
// issue the next prologue after it.
#define G256P_STAGE (2 * G256_BUF)            // byte offset of the wave-private staging area
#define G256P_LDS (2 * G256_BUF + 8 * 4096)   // 160 KiB: all of a CU's LDS

template <int EPI>
__global__ __launch_bounds__(512) void gemm256p_kernel(GemmArgs p, int tiles, int cgw) {
    static_assert(EPI == EPI_GELU_BF16 || EPI == EPI_RELU_BF16 || EPI == EPI_QKV, "bf16 outputs only");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 2, wc = wave & 3, l15 = lane & 15, q = lane >> 4;
    const int tiles_m = (p.M + 255) >> 8;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_round = (gridDim.x >> 3);
    // tile `seq` of the column-group-major order: groups of cgw column tiles, rows inside a group
    const int per_group = tiles_m * cgw;
#define G256P_TILE(SEQ, M0, N0)                                      \
    {                                                                \
        const int g_ = (SEQ) / per_group, r_ = (SEQ) % per_group;    \
        M0 = (r_ / cgw) << 8;                                        \
        N0 = (g_ * cgw + r_ % cgw) << 8;                             \
    }
    // staging pieces and fragment addresses as in g256_mainloop2
    const int srow = tid >> 3;
    const int schunk = (tid & 7) ^ ((tid >> 4) & 7);
    uint32_t aoff00, aoff01, aoff10, aoff11, boff00, boff01, boff10, boff11;
#define G256P_OFFSETS(M0, N0)                                                                    \
    {                                                                                            \
        int r0_ = (M0) + srow, r1_ = (M0) + 128 + srow, r2_ = (M0) + 64 + srow, r3_ = (M0) + 192 + srow; \
        r0_ = r0_ < p.M ? r0_ : p.M - 1; r1_ = r1_ < p.M ? r1_ : p.M - 1;                        \
        r2_ = r2_ < p.M ? r2_ : p.M - 1; r3_ = r3_ < p.M ? r3_ : p.M - 1;                        \
        aoff00 = (uint32_t)(((size_t)r0_ * p.lda + schunk * 8) * 2);                                     \
        aoff01 = (uint32_t)(((size_t)r1_ * p.lda + schunk * 8) * 2);                                     \
        aoff10 = (uint32_t)(((size_t)r2_ * p.lda + schunk * 8) * 2);                                     \
        aoff11 = (uint32_t)(((size_t)r3_ * p.lda + schunk * 8) * 2);                                     \
        const int bn_ = (N0) + (srow >> 5) * 64 + (srow & 31);                                   \
        boff00 = (uint32_t)((((size_t)(bn_)) * p.ldw + schunk * 8) * 2);                                   \
        boff01 = (uint32_t)((((size_t)(bn_ + 128)) * p.ldw + schunk * 8) * 2);                             \
        boff10 = (uint32_t)((((size_t)(bn_ + 32)) * p.ldw + schunk * 8) * 2);                              \
        boff11 = (uint32_t)((((size_t)(bn_ + 160)) * p.ldw + schunk * 8) * 2);                             \
    }
#define G256P_PROLOGUE()                   \
    G256_STAGE_B(0, 0, 0)                  \
    G256_STAGE_B(1, 0, 0)                  \
    G256_STAGE_A(0, 0, 0)                  \
    G256_STAGE_A(1, 0, 0)                  \
    G256_STAGE_B(0, 1, G256_BUF)           \
    G256_STAGE_B(1, 1, G256_BUF)
    const int sw = (lane >> 1) & 7;
    const uint32_t a_k0 = (uint32_t)((wr * 64 + l15) * 128 + ((q ^ sw) << 4));
    const uint32_t b_k0 = (uint32_t)(2 * G256_HALF + (wc * 32 + l15) * 128 + ((q ^ sw) << 4));
    const int nk = p.K >> 6;

    // Which tiles of the column-group-major list a workgroup walks: 32 per XCD and round. (Round 3 measured
    // and dropped the alternative of an XCD OWNING a column group and a quarter of its tiles for the whole
    // launch - W panels resident in its L2, every A panel fetched by one XCD per group: fc1 at 30 streams
    // 105.7 us against 100.5 with this dealing, ViT-L fc1 283 against 286: profiles/r03_tile_order_ab.txt.)
    int seq = xcd * 32 + slot;                 // round 0: chunk xcd
    const int seq_step = 8 * 32;               // next round: chunk + 8
    const int seq_end = tiles;
    (void)per_round;
    int m0 = 0, n0 = 0;
    bool after16 = false;
#ifdef VT_STAMPS
    unsigned long long sp_wait = 0, sp_main = 0, sp_epi = 0, sp_a, sp_b;
    const unsigned long long sp_t0 = __builtin_amdgcn_s_memtime(), sp_r0 = __builtin_amdgcn_s_memrealtime();
    (void)sp_r0;
#define G256P_T(v) { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define G256P_T(v)
#endif
    if (slot < 32 && seq < seq_end) {
        G256P_TILE(seq, m0, n0)
        G256P_OFFSETS(m0, n0)
        G256P_PROLOGUE()
    }
    for (; slot < 32 && seq < seq_end; seq += seq_step) {
        acc256_t acc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int nf = 0; nf < 2; ++nf) acc[i][mf][j][nf] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        bool v_tile = false;
        float scale = 1.0f;
        if constexpr (EPI == EPI_QKV) {
            v_tile = n0 >= 2 * p.D;
            scale = (n0 < p.D) ? ATT_Q_SCALE : 1.0f;
        }
        // ---- this tile's epilogue terms, by LDS-DMA into the wave's own staging area (idle until the
        // epilogue): [0, 1 KiB) the folded LayerNorm's row terms of the wave's 128 rows (float2 each),
        // [1 KiB, 2 KiB) the bias and [2 KiB, 3 KiB) the column sums of its 64 columns (256 B, four
        // copies each: every lane fetches, one source array per instruction - a per-lane choice between
        // two arrays makes hipcc loop over the distinct base pointers, and the lanes it masks off in
        // each round still write their LDS slots). Fetched with ordinary loads at the epilogue's top
        // they cost every tile an exposed round trip to wherever the previous kernel left the row terms
        // (measured: QKV 67 -> 84 us). Up to three more entries at the head of this tile's part of the
        // vector-memory queue: the counted waits of the main loop then wait for up to three operations
        // more than they need (always safe; they still never wait for a piece issued after the one they
        // guard); the data has landed long before the loop's last wait (vmcnt(0)), only this wave reads it.
        const bool ln = p.rowstat != nullptr;      // launch_persistent: M is even when it is set
        {
            int te = tid;
            asm volatile("" : "+v"(te));           // lane-derived addresses rebuilt per tile (see the epilogue)
            const int el = te & 63;
            char* ow = smem + G256P_STAGE + wave * 4096;
            if (ln) {
                int r = m0 + wr * 128 + 2 * el;
                r = r < p.M - 1 ? r : p.M - 2;
                glds16(reinterpret_cast<const char*>(p.rowstat) + (size_t)r * 8, ow);
            }
            glds16(p.bias + n0 + wc * 64 + 4 * (el & 15), ow + 1024);
            if (ln) glds16(p.colsum + n0 + wc * 64 + 4 * (el & 15), ow + 2048);
        }
        // ---- main loop (schedule v2); its prologue was issued before the previous tile's epilogue.
        // The counted waits below assume only LDS-DMA in the queue; stores of the previous epilogue
        // that are still in flight sit between the prologue pieces and this tile's first staging
        // pieces, so the first waits may also wait for them (over-waiting is always safe).
        {
            const char* pa0 = smem + a_k0;
            const char* pa1 = smem + (a_k0 ^ 64);
            const char* pb0 = smem + b_k0;
            const char* pb1 = smem + (b_k0 ^ 64);
            bf16x8_t Af[4][2], Bf[2][2][2];
            // `after16`: the previous epilogue of this wave interleaved its 16 stores with the 12
            // prologue pieces in the order shown there (vmcnt retires loads, stores and LDS-DMA
            // together, in issue order). B(0), A0(0) = the first 6 pieces have 18 younger operations
            // behind them; after the 4 pieces of A(1) are issued, A1(0) has 18 as well: the first two
            // waits of this tile leave the stores in flight instead of waiting for their acknowledgement
            bool first_kt = after16;
            G256P_T(sp_a)
            if (first_kt) wait_vmcnt<18>(); else wait_vmcnt<6>();
            __builtin_amdgcn_s_barrier();
            if (wr == 1) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef VT_STAMPS
            G256P_T(sp_b) sp_wait += sp_b - sp_a;
#endif
            int bo = 0, kt = 0;
#define G256_FLIP()                                                           \
    bo ^= G256_BUF;                                                           \
    pa0 = smem + (a_k0 ^ bo); pa1 = smem + ((a_k0 ^ 64) ^ bo);                \
    pb0 = smem + (b_k0 ^ bo); pb1 = smem + ((b_k0 ^ 64) ^ bo);
#define G256P_BODY(SW)                                                        \
            for (; kt < nk - 2; ++kt) {                                       \
                const int bn = bo ^ G256_BUF;                                 \
                G256_READ_B(0) G256_READ_B(1) G256_READ_A(0)                  \
                G256_STAGE_A(0, kt + 1, bn)                                   \
                G256_STAGE_A(1, kt + 1, bn)                                   \
                if (first_kt) wait_vmcnt<18>(); else wait_vmcnt<8>();         \
                first_kt = false;                                             \
                G256_COMPUTE2(0)                                              \
                G256_READ_A(1)                                                \
                G256_STAGE_B(0, kt + 2, bo)                                   \
                G256_STAGE_B(1, kt + 2, bo)                                   \
                wait_vmcnt<6>();                                              \
                G256_COMPUTE2(1)                                              \
                G256_FLIP()                                                   \
            }                                                                 \
            {                                                                 \
                const int bn = bo ^ G256_BUF;                                 \
                G256_READ_B(0) G256_READ_B(1) G256_READ_A(0)                  \
                G256_STAGE_A(0, kt + 1, bn)                                   \
                G256_STAGE_A(1, kt + 1, bn)                                   \
                if (first_kt) wait_vmcnt<18>(); else wait_vmcnt<8>();         \
                first_kt = false;                                             \
                G256_COMPUTE2(0)                                              \
                G256_READ_A(1)                                                \
                wait_vmcnt<2>();                                              \
                G256_COMPUTE2(1)                                              \
                G256_FLIP()                                                   \
            }                                                                 \
            {                                                                 \
                G256_READ_B(0) G256_READ_B(1) G256_READ_A(0)                  \
                wait_vmcnt<0>();                                              \
                G256_COMPUTE2(0)                                              \
                G256_READ_A(1)                                                \
                G256_COMPUTE2(1)                                              \
            }
            if (!v_tile) {
                constexpr bool SWAP = true;
                G256P_BODY(true)
            } else {
                constexpr bool SWAP = false;
                G256P_BODY(false)
            }
#undef G256_FLIP
            if (wr == 0) __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
#ifdef VT_STAMPS
            G256P_T(sp_a) sp_main += sp_a - sp_b;
#endif
        }
        // every wave has finished reading the ring
        const int tm0 = m0, tn0 = n0;                 // the tile whose accumulators we hold
        const int nseq = seq + seq_step;
        const bool more = nseq < seq_end;
        after16 = false;
        if (more) { G256P_TILE(nseq, m0, n0) }
        if (!v_tile) {
            if (more) { G256P_OFFSETS(m0, n0) }
            // ---- wave-private epilogue, software-pipelined over the wave's eight 16-row blocks
            // (block k = accumulator rows i = k >> 2, mf = k & 3; two 2-KiB halves of this wave's
            // 4 KiB of staging LDS alternate):
            //     bias / activation / pack block k -> ds_write (half k & 1)
            //     global stores of block k-1 (its LDS reads were issued one block ago)
            //     one sixth of the NEXT tile's prologue (2 LDS-DMA pieces; the ring is dead)
            //     ds_read block k
            // so the LDS round trip, the store issue and the LDS-DMA issue back-pressure (each ~100+
            // cycles when issued back to back) sit between VALU work instead of stalling the wave.
            // LDS executes one wave's operations in order: no wait is needed between a wave's own
            // write and read of a half, nor before it overwrites a half it has already read.
            // Lane-derived addresses are rebuilt from an opaque copy of the thread id: hoisted out of
            // the tile loop they would be spilled around the main loop (guide: "a lane-constant address
            // hoisted to kernel entry is spilled around the tile loop - recompute per block").
            int te = tid;
            asm volatile("" : "+v"(te));
            const int el = te & 63, e15 = el & 15, eq = el >> 4;
            // folded LayerNorm (vt_common.hpp): y = a_r * acc + (b_r * colsum[n] + bias[n]); the row terms of
            // the lane's eight rows (one per 16-row block), the bias and the column sums come from the
            // wave's staging area, where the DMA of the tile's top left them (inline-asm LDS reads, one
            // wait naming every destination: compiler-visible reads would be sunk into the pipelined
            // blocks below, each behind a vmcnt(0)). Without a LayerNorm a_r = 1, b_r = 0:
            // fma(1, acc, bias) = acc + bias.
            char* ow = smem + G256P_STAGE + wave * 4096;
            u32x4_t tb[2][2], tc[2][2];
            unsigned long long tr[8];
            {
                const uint32_t a0 = lds_addr(ow) + 1024 + eq * 16, r0 = lds_addr(ow) + e15 * 8;
                lds_read128_asm<0>(tb[0][0], a0); lds_read128_asm<64>(tb[0][1], a0);
                lds_read128_asm<128>(tb[1][0], a0); lds_read128_asm<192>(tb[1][1], a0);
                lds_read128_asm<1024>(tc[0][0], a0); lds_read128_asm<1024 + 64>(tc[0][1], a0);
                lds_read128_asm<1024 + 128>(tc[1][0], a0); lds_read128_asm<1024 + 192>(tc[1][1], a0);
                lds_read64_asm<0>(tr[0], r0); lds_read64_asm<128>(tr[1], r0); lds_read64_asm<256>(tr[2], r0);
                lds_read64_asm<384>(tr[3], r0); lds_read64_asm<512>(tr[4], r0); lds_read64_asm<640>(tr[5], r0);
                lds_read64_asm<768>(tr[6], r0); lds_read64_asm<896>(tr[7], r0);
                asm volatile("s_waitcnt lgkmcnt(0)"
                             : "+v"(tb[0][0]), "+v"(tb[0][1]), "+v"(tb[1][0]), "+v"(tb[1][1]), "+v"(tc[0][0]), "+v"(tc[0][1]),
                               "+v"(tc[1][0]), "+v"(tc[1][1]), "+v"(tr[0]), "+v"(tr[1]), "+v"(tr[2]), "+v"(tr[3]), "+v"(tr[4]),
                               "+v"(tr[5]), "+v"(tr[6]), "+v"(tr[7])
                             :
                             : "memory");
            }
            f32x4_t bias4[2][2], cs4[2][2];
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    bias4[j][nf] = __builtin_bit_cast(f32x4_t, tb[j][nf]);
                    cs4[j][nf] = ln ? __builtin_bit_cast(f32x4_t, tc[j][nf]) : f32x4_t{0.f, 0.f, 0.f, 0.f};
                }
            float2 rs8[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                rs8[k] = ln ? make_float2(__uint_as_float((uint32_t)tr[k]), __uint_as_float((uint32_t)(tr[k] >> 32)))
                            : make_float2(1.0f, 0.0f);
            const int rr = el >> 3, rc = el & 7;       // read-out: 8 rows per instruction, 16-B chunk rc
            bf16_t* const obase = (EPI == EPI_QKV)
                                      ? p.qk + (size_t)(tm0 + wr * 128 + rr) * (2 * p.D) + tn0 + wc * 64 + rc * 8
                                      : p.Cb + (size_t)(tm0 + wr * 128 + rr) * p.ldcb + tn0 + wc * 64 + rc * 8;
            const size_t ostride8 = 8 * ((EPI == EPI_QKV) ? (size_t)(2 * p.D) : (size_t)p.ldcb);
            const int rows_left = p.M - (tm0 + wr * 128);   // rows of this wave's piece inside M (wave-uniform)
            const bool full = rows_left >= 128;
            const int wsw = (e15 & 7) << 1;
            const uint32_t wb_ = lds_addr(ow) + e15 * 128;
            // 8-B piece c8 = j*8 + nf*4 + eq of row e15, stored at c8 ^ wsw
            const uint32_t wa00 = wb_ + (((0 + eq) ^ wsw) << 3), wa01 = wb_ + (((4 + eq) ^ wsw) << 3);
            const uint32_t wa10 = wb_ + (((8 + eq) ^ wsw) << 3), wa11 = wb_ + (((12 + eq) ^ wsw) << 3);
            const uint32_t ra = lds_addr(ow) + rr * 128 + ((rc ^ rr) << 4);
            u32x4_t o0, o1;
// Output stores of the persistent kernel are WRITE-THROUGH (sc1): a plain store keeps its line in the XCD's
// L2 (MI355X guide, "stores of each flavour"), and a round's 32 output tiles per XCD are 4.2 MB - as much as the
// round's operand tiles - pushed through a 4-MiB L2 that the 32 CUs are re-reading those operands from.
// Round 4, alternating builds in one process (profiles/r04_fc1_tile_order_cache_policy.txt): fc1 100.5 -> 96.3 us,
// QKV 70.8 -> 69.4 us, ViT-L fc1 281.5 -> 277.2 us; fabric reads 151.7 -> 145.5 MB.
#define G256P_ST16(PTR, V) asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(PTR), "v"(V) : "memory")
#define G256P_STORE(KB)                                                                           \
    lds_wait2_asm(o0, o1);                                                                       \
    if (full) {                                                                                  \
        G256P_ST16(obase + (size_t)((KB) * 2) * ostride8, o0);                                   \
        G256P_ST16(obase + (size_t)((KB) * 2 + 1) * ostride8, o1);                               \
    } else {                                                                                     \
        if ((KB) * 16 + rr < rows_left) *reinterpret_cast<u32x4_t*>(obase + (size_t)((KB) * 2) * ostride8) = o0; \
        if ((KB) * 16 + 8 + rr < rows_left) *reinterpret_cast<u32x4_t*>(obase + (size_t)((KB) * 2 + 1) * ostride8) = o1; \
    }
#define VT_GELU2X4(V) gelu_erf2x4(V)
// the four 2 x 2-element groups of a block are computed FIRST and written to LDS afterwards: an asm statement is
// a scheduling fence, and with one ds_write per group hipcc had only that group's two dependent chains to
// interleave - 218 s_nop (packed-fma result -> next packed fma, v_exp -> use) per tile and wave in a section
// that is bound by vector issue. With eight independent chains per block the hazards are covered by real work.
#define G256P_EPI_X(K, J, NF, E)                                                                 \
    __builtin_elementwise_fma(f32v2_t{rs8[K].x, rs8[K].x},                                       \
        f32v2_t{acc[(K) >> 2][(K) & 3][J][NF][2 * (E)], acc[(K) >> 2][(K) & 3][J][NF][2 * (E) + 1]}, \
        __builtin_elementwise_fma(f32v2_t{rs8[K].y, rs8[K].y}, f32v2_t{cs4[J][NF][2 * (E)], cs4[J][NF][2 * (E) + 1]}, \
                                  f32v2_t{bias4[J][NF][2 * (E)], bias4[J][NF][2 * (E) + 1]}))
#define G256P_EPI_CALC2(K, J, LO0, HI0, LO1, HI1)                                                \
    {                                                                                            \
        f32v2_t v[4] = {G256P_EPI_X(K, J, 0, 0), G256P_EPI_X(K, J, 0, 1), G256P_EPI_X(K, J, 1, 0), G256P_EPI_X(K, J, 1, 1)}; \
        if constexpr (EPI == EPI_GELU_BF16) {                                                    \
            VT_GELU2X4(v);                                                                       \
        } else if constexpr (EPI == EPI_RELU_BF16) {                                             \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = f32v2_t{fmaxf(v[e].x, 0.0f), fmaxf(v[e].y, 0.0f)}; \
        } else {                                                                                 \
            _Pragma("unroll") for (int e = 0; e < 4; ++e) v[e] = v[e] * scale;                    \
        }                                                                                        \
        LO0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[0], bf16v2_t));             \
        HI0 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[1], bf16v2_t));             \
        LO1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[2], bf16v2_t));             \
        HI1 = __builtin_bit_cast(uint32_t, __builtin_convertvector(v[3], bf16v2_t));             \
    }
#define G256P_EPI_BLOCK(K, STAGE_STMT)                                                           \
    {                                                                                            \
        uint32_t l00, h00, l01, h01, l10, h10, l11, h11;                                         \
        G256P_EPI_CALC2(K, 0, l00, h00, l01, h01)                                                \
        G256P_EPI_CALC2(K, 1, l10, h10, l11, h11)                                                \
        lds_write64_asm<((K) & 1) * 2048>(wa00, l00, h00);                                       \
        lds_write64_asm<((K) & 1) * 2048>(wa01, l01, h01);                                       \
        lds_write64_asm<((K) & 1) * 2048>(wa10, l10, h10);                                       \
        lds_write64_asm<((K) & 1) * 2048>(wa11, l11, h11);                                       \
        if ((K) > 0) { G256P_STORE((K) - 1) }                                                    \
        if (more) { STAGE_STMT }                                                                 \
        lds_read128_asm<((K) & 1) * 2048>(o0, ra);                                               \
        lds_read128_asm<((K) & 1) * 2048 + 1024>(o1, ra);                                        \
    }
            G256P_EPI_BLOCK(0, G256_STAGE_B(0, 0, 0))
            G256P_EPI_BLOCK(1, G256_STAGE_B(1, 0, 0))
            G256P_EPI_BLOCK(2, G256_STAGE_A(0, 0, 0))
            G256P_EPI_BLOCK(3, G256_STAGE_A(1, 0, 0))
            G256P_EPI_BLOCK(4, G256_STAGE_B(0, 1, G256_BUF))
            G256P_EPI_BLOCK(5, G256_STAGE_B(1, 1, G256_BUF))
            G256P_EPI_BLOCK(6, )
            G256P_EPI_BLOCK(7, )
            G256P_STORE(7)
#undef G256P_EPI_BLOCK
#undef G256P_EPI_CALC2
#undef G256P_EPI_X
#undef G256P_STORE
#undef G256P_ST16
            // queue of this wave now: g g | s s g g | s s g g | s s g g | s s g g | s s g g | s s | s s | s s
            after16 = more && full;
        } else if constexpr (EPI == EPI_QKV) {
            // V tile: block-wide transpose through the (dead) ring, then the next prologue
            const int heads = p.D >> 6;
            int tv = tid;
            asm volatile("" : "+v"(tv));               // addresses rebuilt per tile (see above)
            const int l15 = tv & 15, q = (tv & 63) >> 4, tid = tv;
            // the tile's epilogue terms in this wave's staging area (DMA of the tile's top; landed before the
            // main loop's last wait): floats [0, 256) row terms, [256, 320) bias, [512, 576) column sums
            const float* sterm = reinterpret_cast<const float*>(smem + G256P_STAGE + wave * 4096);
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int nf = 0; nf < 2; ++nf) {
                    const int drow = wc * 64 + j * 32 + nf * 16 + l15;
                    const float bias = sterm[256 + j * 32 + nf * 16 + l15];
                    const float cs = ln ? sterm[512 + j * 32 + nf * 16 + l15] : 0.0f;
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf) {
                            const int qp = (p.vt_perm && !(p.tokens & 15)) ? ((q & 1) << 1 | (q >> 1)) : q;
                            const int c8 = wr * 32 + i * 16 + mf * 4 + qp;
                            const f32x4_t a = acc[i][mf][j][nf];
                            // row terms of tokens 4q .. 4q+3 of this 16-row block: 32 B of the wave's table
                            f32x4_t r01 = {1.f, 0.f, 1.f, 0.f}, r23 = r01;
                            if (ln) {
                                r01 = *reinterpret_cast<const f32x4_t*>(sterm + (i * 64 + mf * 16 + 4 * q) * 2);
                                r23 = *reinterpret_cast<const f32x4_t*>(sterm + (i * 64 + mf * 16 + 4 * q) * 2 + 4);
                            }
                            const float y0 = __builtin_fmaf(r01[0], a[0], __builtin_fmaf(r01[1], cs, bias));
                            const float y1 = __builtin_fmaf(r01[2], a[1], __builtin_fmaf(r01[3], cs, bias));
                            const float y2 = __builtin_fmaf(r23[0], a[2], __builtin_fmaf(r23[1], cs, bias));
                            const float y3 = __builtin_fmaf(r23[2], a[3], __builtin_fmaf(r23[3], cs, bias));
                            *reinterpret_cast<uint2*>(smem + drow * 512 + ((c8 ^ ((drow & 7) << 1)) << 3)) =
                                make_uint2(pack_bf16x2(y0, y1), pack_bf16x2(y2, y3));
                        }
                }
            __syncthreads();
            if (p.vt_perm && (p.tokens & 15)) {
                const int c8r = tid & 63;
                for (int it = 0; it < 32; ++it) {
                    const int r = it * 8 + (tid >> 6);
                    const int m = tm0 + c8r * 4, nv = tn0 + r - 2 * p.D;
                    if (m >= p.M) continue;
                    const int b = m / p.tokens, t = m % p.tokens;
                    bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + attn_perm16(t);
                    *reinterpret_cast<uint2*>(dst) =
                        *reinterpret_cast<const uint2*>(smem + r * 512 + ((c8r ^ ((r & 7) << 1)) << 3));
                }
            } else {
                const int c16 = tid & 31;
                for (int it = 0; it < 16; ++it) {
                    const int r = it * 16 + (tid >> 5);
                    const int m = tm0 + c16 * 8, nv = tn0 + r - 2 * p.D;
                    if (m >= p.M) continue;
                    const int b = m / p.tokens, t = m % p.tokens;
                    bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + t;
                    const char* src = smem + r * 512 + ((c16 ^ (r & 7)) << 4);
                    if (t + 8 <= p.tokens && m + 8 <= p.M && ((t & 7) == 0)) {
                        *reinterpret_cast<uint4*>(dst) = *reinterpret_cast<const uint4*>(src);
                    } else {
                        for (int e = 0; e < 8 && m + e < p.M; ++e) {
                            const int me = m + e, be = me / p.tokens, te = me % p.tokens;
                            p.vt[((size_t)(be * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + te] =
                                *reinterpret_cast<const bf16_t*>(src + 2 * e);
                        }
                    }
                }
            }
            __syncthreads();                              // ring free again
            if (more) {
                G256P_OFFSETS(m0, n0)
                G256P_PROLOGUE()
            }
        }
#ifdef VT_STAMPS
        G256P_T(sp_b) sp_epi += sp_b - sp_a;
#endif
    }
#ifdef VT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * 8 + wave) * 4;
        d[0] = sp_wait; d[1] = sp_epi; d[2] = sp_main; d[3] = __builtin_amdgcn_s_memtime() - sp_t0;
#ifdef VT_STAMPS_CLOCK
        d[0] = __builtin_amdgcn_s_memrealtime() - sp_r0;
#endif
    }
#endif
#undef G256P_BODY
#undef G256P_TILE
#undef G256P_OFFSETS
#undef G256P_PROLOGUE
}

// column-group width for the persistent tile order: the divisor c of tiles_n minimising 32/c + c
static int g256p_cgw(int tiles_n) {
    int best = 1;
    double cost = 1e9;
    for (int c = 1; c <= tiles_n; ++c)
        if (tiles_n % c == 0) {
            const double k = 32.0 / c + c;
            if (k < cost - 1e-9) { cost = k; best = c; }
        }
    return best;
}

template <int EPI>
hipError_t prepare_persistent() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256p_kernel<EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, G256P_LDS);
}

template <int EPI>
hipError_t launch_persistent(const GemmArgs& a, hipStream_t st) {
    const int tiles_n = a.N / 256, tiles = ((a.M + 255) / 256) * tiles_n;
    vt_launch((gemm256p_kernel<EPI>), dim3(256), dim3(512), G256P_LDS, st, a, tiles,
                       g256p_cgw(tiles_n));
    return hipGetLastError();
}

template <int EPI>
hipError_t prepare_one() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256_kernel<EPI>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, G256_LDS);
}

template <int EPI>
hipError_t launch_one(const GemmArgs& a, hipStream_t st) {
    const int tiles = ((a.M + 255) / 256) * (a.N / 256);
    vt_launch((gemm256_kernel<EPI>), dim3(tiles), dim3(512), G256_LDS, st, a);
    return hipGetLastError();
}

}  // namespace

#ifndef VT_TU_SHA256
#define VT_TU_SHA256 "unstamped"
#endif
const char* gemm256_build_id() { return VT_TU_SHA256; }

hipError_t gemm256_prepare() {
    hipError_t e;
    if ((e = prepare_one<EPI_F32_POS>()) != hipSuccess) return e;
    if ((e = prepare_one<EPI_RESID>()) != hipSuccess) return e;
    if ((e = prepare_one<EPI_GELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_one<EPI_RELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_one<EPI_QKV>()) != hipSuccess) return e;
    if ((e = prepare_persistent<EPI_GELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_persistent<EPI_RELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_persistent<EPI_QKV>()) != hipSuccess) return e;
    return prepare_one<EPI_F32>();
}

// does the shape fit the 256x256 kernels? (launch_gemm takes the 4-wave kernel of k_gemm.hip otherwise)
bool gemm256_fits(const GemmArgs& a, int epilogue) {
    if (a.M <= 0 || a.N % 256 != 0 || a.K % 64 != 0 || a.K < 128) return false;
    // unsigned 32-bit byte offsets from the operand base pointers (+ the K offset added to the base):
    // vt_plan_engines keeps every engine below this (VT_GEMM256_MAX_OPERAND_BYTES)
    if ((long long)a.M * a.lda * 2 >= VT_GEMM256_MAX_OPERAND_BYTES || (long long)a.N * a.ldw * 2 >= VT_GEMM256_MAX_OPERAND_BYTES)
        return false;
    if ((a.lda & 7) || (a.ldw & 7) || a.conv_grid > 0) return false;
    switch (epilogue) {
        case EPI_F32_POS:
        case EPI_RESID:
        case EPI_F32:
            return !((a.ldx & 7) || !a.Xh || !a.Xl || (epilogue == EPI_F32_POS && (!a.pos || a.pos_rows < 1)));
        case EPI_GELU_BF16:
        case EPI_RELU_BF16:
            return !((a.ldcb & 7) || !a.Cb || !a.bias || (a.rowstat && !a.colsum));
        case EPI_QKV:
            return !(a.D % 256 != 0 || a.N != 3 * a.D || (a.tokens & 3) || (a.npad & 3) || !a.bias ||
                     (a.rowstat && !a.colsum));
        default: return false;
    }
}

// hipErrorInvalidValue: the shape does not fit this kernel
hipError_t launch_gemm256(const GemmArgs& a, int epilogue, bool persistent, hipStream_t st) {
    if (!gemm256_fits(a, epilogue)) return hipErrorInvalidValue;
    if (persistent) {   // bf16 outputs, more tiles than CUs; anything else takes the one-tile-per-workgroup form
        const int tiles = ((a.M + 255) / 256) * (a.N / 256);
        if (tiles > 256 && !(a.rowstat && (a.M & 1)))     // the row terms are fetched two rows per lane (16-B LDS-DMA)
            switch (epilogue) {
                case EPI_GELU_BF16: return launch_persistent<EPI_GELU_BF16>(a, st);
                case EPI_RELU_BF16: return launch_persistent<EPI_RELU_BF16>(a, st);
                case EPI_QKV: return launch_persistent<EPI_QKV>(a, st);
                default: break;
            }
    }
    switch (epilogue) {
        case EPI_F32_POS: return launch_one<EPI_F32_POS>(a, st);
        case EPI_RESID: return launch_one<EPI_RESID>(a, st);
        case EPI_GELU_BF16: return launch_one<EPI_GELU_BF16>(a, st);
        case EPI_RELU_BF16: return launch_one<EPI_RELU_BF16>(a, st);
        case EPI_QKV: return launch_one<EPI_QKV>(a, st);
        default: return launch_one<EPI_F32>(a, st);
    }
}
