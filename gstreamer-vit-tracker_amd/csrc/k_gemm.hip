// k_gemm.hip — bf16 MFMA GEMM for gfx950: C[M,N] = A[M,K] · W[N,K]^T (+ fused epilogue).
//
// Every dense contraction of the tracker goes through this kernel: patch-embed, QKV, attention
// output projection, MLP fc1/fc2, and the head's 1x1 / 3x3 (implicit-GEMM) convolutions. Both operands
// are K-contiguous ("A rows" and "W rows"), so one staging routine and one fragment reader serve
// both.
//
// Structure (CDNA4):
//   * 256 threads = 4 waves in a 2x2 arrangement; block tile BM x BN (128x128 or 64x64), BK = 64.
//   * global -> LDS by LDS-DMA (global_load_lds_dwordx4, 16 B per lane, 1 KiB per wave
//     instruction = 8 tile rows of 128 B) into a ring of NS LDS stages: the loads of K-tiles
//     t+1 .. t+NS-2 stay in flight across the barrier while the MFMAs of tile t run (counted
//     s_waitcnt vmcnt(N), raw s_barrier: a plain __syncthreads() would drain the queue). At
//     M = 720 a block's MFMA work per K-tile (~0.1 us) is far shorter than an L2/HBM round trip,
//     so the prefetch distance, not the tile shape, sets the time per K-tile.
//   * LDS image: rows of 128 B (64 bf16); the 16-B chunk c of row r is stored at chunk
//     c ^ ((r >> 1) & 7). LDS-DMA writes lane-linear, so the XOR is applied to the per-lane
//     SOURCE address and again on the read (guide rule 21). With this swizzle the ds_read_b128
//     of a 32-row MFMA operand is bank-conflict free for both 16-lane halves of each read group.
//   * v_mfma_f32_32x32x16_bf16, f32 accumulate. The MFMA operand order is chosen per epilogue so
//     that the store side is wide: f32 outputs keep the column on the lane (128 B contiguous per
//     row per store), bf16 row-major outputs put the ROW on the lane so each lane owns 4
//     consecutive columns per accumulator quad (one 8-B store).
#include "vt_common.hpp"
#include "k_gemm_util.hpp"

#define GEMM_BK 64
#define ROW_BYTES 128

// LDS ring: NS stages of (BM + BN) rows x 128 B (K-tile depth 64)
__host__ __device__ constexpr int ring_bytes(int ns, int bm, int bn, int bk = GEMM_BK) { return ns * (bm + bn) * bk * 2; }

// accumulator element `reg` of a 32x32 MFMA tile: row offset inside the tile
__device__ __forceinline__ int acc_row(int reg, int half) {
    return (reg & 3) + 8 * (reg >> 2) + 4 * half;
}

template <int BM, int BN, int WVM, int WVN, int NS, int BK, bool ROW_ON_LANE, bool CONV = false>
__device__ __forceinline__ void gemm_mainloop(const GemmArgs& p, char* smem, int m0, int n0,
                                              f32x16_t (&acc)[BM / WVM / 32][BN / WVN / 32]) {
    static_assert(BK == 64 || BK == 128, "K-tile depth");
    constexpr int ROWB = BK * 2;                // bytes per LDS row (128 or 256)
    constexpr int CPR = BK / 8;                 // 16-B chunks per row (8 or 16)
    constexpr int RPP = 1024 / ROWB;            // rows per 1-KiB LDS-DMA piece (8 or 4)
    constexpr int NW = WVM * WVN;               // waves per block
    constexpr int WM = BM / WVM, WN = BN / WVN; // wave tile
    constexpr int TM = WM / 32, TN = WN / 32;   // 32x32 MFMA tiles per wave in each direction
    constexpr int PA = BM / (RPP * NW), PB = BN / (RPP * NW);  // 1-KiB LDS-DMA pieces per wave
    static_assert(PA >= 1 && PB >= 1 && TM >= 1 && TN >= 1, "tile too small for the wave grid");
    constexpr int STAGE = (BM + BN) * ROWB;
    // chunk swizzle: the ds_read_b128 of a 32-row MFMA operand must hit 16 distinct 16-B slots (one
    // 256-B bank line) per 16-lane group. 128-B rows (BK 64): two rows share a line, c ^ ((r >> 1) & 7);
    // 256-B rows (BK 128): one row per line, c ^ (r & 15)
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : row & 15; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave / WVN, wc = wave % WVN;
    const int l31 = lane & 31, half = lane >> 5;

    // per-lane source pointers for the LDS-DMA pieces this wave issues (rows fixed over K)
    const bf16_t* asrc[PA];
    const bf16_t* bsrc[PB];
    // implicit 3x3 convolution: the row's cell in its map and the column chunk this lane fetches
    int cv_cell[PA], cv_y[PA], cv_x[PA], cv_c[PA];
    const bool conv = CONV && p.conv_grid > 0;
#pragma unroll
    for (int j = 0; j < PA; ++j) {
        const int row = (wave * PA + j) * RPP + (lane / CPR);
        const int c = (lane % CPR) ^ swz(row);
        int gm = m0 + row;
        gm = gm < p.M ? gm : p.M - 1;
        asrc[j] = p.A + (size_t)gm * p.lda + c * 8;
        if constexpr (CONV) {
            const int g = p.conv_grid > 0 ? p.conv_grid : 1, cell = gm % (g * g);
            cv_cell[j] = gm; cv_y[j] = cell / g; cv_x[j] = cell % g; cv_c[j] = c * 8;
        }
    }
#pragma unroll
    for (int j = 0; j < PB; ++j) {
        const int row = (wave * PB + j) * RPP + (lane / CPR);
        const int c = (lane % CPR) ^ swz(row);
        bsrc[j] = p.W + (size_t)(n0 + row) * p.ldw + c * 8;
    }
    // piece q of a stage: q < PA -> A rows, else W rows (each piece = one global_load_lds, 1 KiB)
    auto issue_piece = [&](int kt, int buf, int q) {
        char* sA = smem + buf * STAGE;
        if (q < PA) {
            if (conv) {       // K-tile kt = 64 channels [cc, cc + 64) of tap (ky, kx) of the 3x3 window
                const int k0 = kt * BK, tap = k0 / p.conv_C, cc = k0 - tap * p.conv_C;
                const int ky = tap / 3, kx = tap - 3 * ky;
                const int y = cv_y[q] + ky - 1, x = cv_x[q] + kx - 1;
                const bool in = y >= 0 && y < p.conv_grid && x >= 0 && x < p.conv_grid;
                const bf16_t* src = in ? p.A + (size_t)(cv_cell[q] + (ky - 1) * p.conv_grid + (kx - 1)) * p.lda + cc + cv_c[q]
                                       : p.zeros + cv_c[q];
                glds16(src, sA + (wave * PA + q) * 1024);
            } else {
                glds16(asrc[q] + kt * BK, sA + (wave * PA + q) * 1024);
            }
        } else {
            glds16(bsrc[q - PA] + kt * BK, sA + BM * ROWB + (wave * PB + (q - PA)) * 1024);
        }
    };
    auto stage = [&](int kt, int buf) {
#pragma unroll
        for (int q = 0; q < PA + PB; ++q) issue_piece(kt, buf, q);
    };

    // fragment read offsets (bytes inside a stage), per k-step the chunk index changes by 2
    int aoff[TM], boff[TN], aswz[TM], bswz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wr * WM + i * 32 + l31;
        aoff[i] = row * ROWB;
        aswz[i] = swz(row);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 32 + l31;
        boff[j] = BM * ROWB + row * ROWB;
        bswz[j] = swz(row);
    }

    constexpr int IPS = PA + PB;  // LDS-DMA instructions per wave per stage
    const int nk = p.K / BK;
    constexpr int KS = BK / 16;                 // MFMA k-steps per stage
#pragma unroll
    for (int s = 0; s < NS - 1; ++s)
        if (s < nk) stage(s, s);
    int cur = 0;
#ifdef VT_STAMPS
    unsigned long long st_wait = 0, st_issue = 0, st_comp = 0, st_t0 = clock64(), st_a, st_b;
#define STAMP(v) { __builtin_amdgcn_sched_barrier(0); v = clock64(); __builtin_amdgcn_sched_barrier(0); }
#else
#define STAMP(v)
#endif
    for (int kt = 0; kt < nk; ++kt) {
        STAMP(st_a)
        // Tile kt must have landed; tiles kt+1 .. kt+NS-2 (those that exist) may stay in flight.
        const int ahead = min(NS - 2, nk - 1 - kt);
        if (NS >= 4 && ahead >= 2) wait_vmcnt<2 * IPS>();
        else if (NS >= 3 && ahead == 1) wait_vmcnt<IPS>();
        else wait_vmcnt<0>();
        // every wave's pieces of tile kt are in LDS, and every wave has finished reading the
        // stage of tile kt-1, which the next prefetch overwrites
        __builtin_amdgcn_s_barrier();
#ifdef VT_STAMPS
        STAMP(st_b) st_wait += st_b - st_a;
#endif
        // The loads of the next stage are NOT issued in one burst here: one global_load_lds costs
        // the issuing wave ~100 cycles of back-pressure from the CU's vector-memory pipeline
        // (measured with in-kernel stamps: 8 pieces = 850 cycles, as long as the 16 MFMAs of the
        // tile). They are spread over the k-steps below, a few after each MFMA group, so that one
        // wave's issue stalls overlap other waves' MFMAs instead of all waves stalling together.
        const bool prefetch = kt + NS - 1 < nk;
        const int pf_kt = kt + NS - 1, pf_buf = cur == 0 ? NS - 1 : cur - 1;
        (void)pf_kt; (void)pf_buf;
#ifdef VT_STAMPS
        STAMP(st_a) st_issue += st_a - st_b;
#endif
        const char* sbase = smem + cur * STAGE;
        // k-steps software-pipelined inside the wave: the fragments of step ks+1 are read from LDS
        // while the MFMAs of step ks issue (the compiler then waits with a counted lgkmcnt instead
        // of draining every read before every MFMA group)
        bf16x8_t af[2][TM], bfr[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[0][i] = *reinterpret_cast<const bf16x8_t*>(sbase + aoff[i] + ((half ^ aswz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bfr[0][j] = *reinterpret_cast<const bf16x8_t*>(sbase + boff[j] + ((half ^ bswz[j]) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            if (ks < KS - 1) {
                const int c = 2 * (ks + 1) + half;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[nb][i] = *reinterpret_cast<const bf16x8_t*>(sbase + aoff[i] +
                                                                   ((c ^ aswz[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[nb][j] = *reinterpret_cast<const bf16x8_t*>(sbase + boff[j] +
                                                                    ((c ^ bswz[j]) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ROW_ON_LANE)  // D[n][m]: lane = m (row of C), registers = n
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cb][j], af[cb][i],
                                                                            acc[i][j], 0, 0, 0);
                    else              // D[m][n]: lane = n (column of C), registers = m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][i], bfr[cb][j],
                                                                            acc[i][j], 0, 0, 0);
                }
            if (prefetch) {
                constexpr int Q0 = 0;  // pieces [ks*IPS/KS, (ks+1)*IPS/KS) go with k-step ks
#pragma unroll
                for (int q = Q0 + ks * IPS / KS; q < (ks + 1) * IPS / KS; ++q)
                    issue_piece(pf_kt, pf_buf, q);
            }
        }
#ifdef VT_STAMPS
        STAMP(st_b) st_comp += st_b - st_a;
#endif
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
#ifdef VT_STAMPS
    if (p.dbg && lane == 0) {
        unsigned long long* d = p.dbg + ((size_t)blockIdx.x * NW + wave) * 4;
        d[0] = st_wait; d[1] = st_issue; d[2] = st_comp; d[3] = clock64() - st_t0;
    }
#endif
}

// ---- loader-wave variant of the main loop (round 5; configs 7-10, small batches) ------------------------------------
// 512 threads: waves 0..3 compute exactly as above (same LDS image, same fragment reads, same MFMA order: bit-identical
// accumulators), waves 4..7 issue every LDS-DMA piece. At one stream a K-step of the 4-wave kernel is 4 (depth 64) or 8
// (depth 128) LDS-DMA instructions per wave at ~100 cycles of issue back-pressure each, a barrier and 4-8 MFMAs - the
// issue stalls, in the computing waves' own instruction streams, were most of it (profiles/r04_deep_ring_small_batch.txt:
// 0.45-0.55 us per K-step whatever the prefetch distance). Here they run beside the MFMA streams, and the ring can be
// deep (the loaders run ahead; a stage is refilled one barrier after its last read).
//   loader, per K-tile:  counted vmcnt (tile kt landed; up to NS - 2 younger tiles stay in flight) | barrier | issue kt + NS - 1
//   computing wave:      barrier | fragment reads + MFMAs of tile kt
// One raw s_barrier per K-tile for all eight waves.
// The loader waves branch off at the top of the kernel, BEFORE the computing waves' hand-waited inline-asm loads (on a
// path where those loads' values are dead hipcc would be free to reuse their destination registers while the loads are
// still in flight), run gemm_lw_loader and leave through the epilogue's barriers.
template <int BM, int BN, int NS, int BK, bool CONV>
__device__ __forceinline__ void gemm_lw_loader(const GemmArgs& p, char* smem, int m0, int n0) {
    static_assert(BK == 64 || BK == 128, "K-tile depth");
    static_assert(NS >= 3, "a ring of at least three stages");
    constexpr int ROWB = BK * 2, CPR = BK / 8, RPP = 1024 / ROWB;
    constexpr int PA = BM / (RPP * 4), PB = BN / (RPP * 4);      // 1-KiB LDS-DMA pieces per LOADER wave and stage
    static_assert(PA >= 1 && PB >= 1, "tile too small for four loader waves");
    constexpr int STAGE = (BM + BN) * ROWB, IPS = PA + PB;
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : row & 15; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK;
    {
        // ---- loader wave lw: pieces lw * PA + j of the A rows, lw * PB + j of the W rows --------------------------
        const int lw = wave - 4;
        const bf16_t* asrc[PA];
        const bf16_t* bsrc[PB];
        int cv_cell[PA], cv_y[PA], cv_x[PA], cv_c[PA];
        const bool conv = CONV && p.conv_grid > 0;
#pragma unroll
        for (int j = 0; j < PA; ++j) {
            const int row = (lw * PA + j) * RPP + (lane / CPR);
            const int c = (lane % CPR) ^ swz(row);
            int gm = m0 + row;
            gm = gm < p.M ? gm : p.M - 1;
            asrc[j] = p.A + (size_t)gm * p.lda + c * 8;
            if constexpr (CONV) {
                const int g = p.conv_grid > 0 ? p.conv_grid : 1, cell = gm % (g * g);
                cv_cell[j] = gm; cv_y[j] = cell / g; cv_x[j] = cell % g; cv_c[j] = c * 8;
            }
        }
#pragma unroll
        for (int j = 0; j < PB; ++j) {
            const int row = (lw * PB + j) * RPP + (lane / CPR);
            const int c = (lane % CPR) ^ swz(row);
            bsrc[j] = p.W + (size_t)(n0 + row) * p.ldw + c * 8;
        }
        auto stage = [&](int kt, int buf) {
            char* sA = smem + buf * STAGE;
#pragma unroll
            for (int q = 0; q < PA; ++q) {
                if (conv) {       // K-tile kt = channels [cc, cc + BK) of tap (ky, kx) of the 3x3 window
                    const int k0 = kt * BK, tap = k0 / p.conv_C, cc = k0 - tap * p.conv_C;
                    const int ky = tap / 3, kx = tap - 3 * ky;
                    const int y = cv_y[q] + ky - 1, x = cv_x[q] + kx - 1;
                    const bool in = y >= 0 && y < p.conv_grid && x >= 0 && x < p.conv_grid;
                    const bf16_t* src = in ? p.A + (size_t)(cv_cell[q] + (ky - 1) * p.conv_grid + (kx - 1)) * p.lda + cc + cv_c[q]
                                           : p.zeros + cv_c[q];
                    glds16(src, sA + (lw * PA + q) * 1024);
                } else {
                    glds16(asrc[q] + kt * BK, sA + (lw * PA + q) * 1024);
                }
            }
#pragma unroll
            for (int q = 0; q < PB; ++q) glds16(bsrc[q] + kt * BK, sA + BM * ROWB + (lw * PB + q) * 1024);
        };
#pragma unroll
        for (int s_ = 0; s_ < NS - 1; ++s_)
            if (s_ < nk) stage(s_, s_);
        int cur = 0;
        for (int kt = 0; kt < nk; ++kt) {
            const int ahead = min(NS - 2, nk - 1 - kt);        // younger tiles that may stay in flight
            switch (ahead) {                                    // s_waitcnt takes an immediate
                case 0: wait_vmcnt<0>(); break;
                case 1: wait_vmcnt<IPS>(); break;
                case 2: wait_vmcnt<2 * IPS>(); break;
                case 3: wait_vmcnt<(NS >= 5 ? 3 : 0) * IPS>(); break;
                default: wait_vmcnt<(NS >= 6 ? 4 : 0) * IPS>(); break;
            }
            __builtin_amdgcn_s_barrier();       // tile kt has landed; the computing waves are done with tile kt - 1
            __builtin_amdgcn_sched_barrier(0);
            if (kt + NS - 1 < nk) stage(kt + NS - 1, cur == 0 ? NS - 1 : cur - 1);
            cur = (cur + 1 == NS) ? 0 : cur + 1;
        }
    }
}

// ---- computing waves: as gemm_mainloop, without the loads ------------------------------------------------------------
template <int BM, int BN, int WVM, int WVN, int NS, int BK, bool ROW_ON_LANE>
__device__ __forceinline__ void gemm_lw_compute(const GemmArgs& p, char* smem,
                                                f32x16_t (&acc)[BM / WVM / 32][BN / WVN / 32]) {
    static_assert(WVM * WVN == 4, "four computing waves beside four loader waves");
    constexpr int ROWB = BK * 2;
    constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;
    static_assert(TM >= 1 && TN >= 1, "tile too small for the wave grid");
    constexpr int STAGE = (BM + BN) * ROWB, KS = BK / 16;
    auto swz = [](int row) { return BK == 64 ? (row >> 1) & 7 : row & 15; };
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nk = p.K / BK;
    const int wr = wave / WVN, wc = wave % WVN;
    const int l31 = lane & 31, half = lane >> 5;
    int aoff[TM], boff[TN], aswz[TM], bswz[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int row = wr * WM + i * 32 + l31;
        aoff[i] = row * ROWB;
        aswz[i] = swz(row);
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int row = wc * WN + j * 32 + l31;
        boff[j] = BM * ROWB + row * ROWB;
        bswz[j] = swz(row);
    }
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const char* sbase = smem + cur * STAGE;
        bf16x8_t af[2][TM], bfr[2][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
            af[0][i] = *reinterpret_cast<const bf16x8_t*>(sbase + aoff[i] + ((half ^ aswz[i]) << 4));
#pragma unroll
        for (int j = 0; j < TN; ++j)
            bfr[0][j] = *reinterpret_cast<const bf16x8_t*>(sbase + boff[j] + ((half ^ bswz[j]) << 4));
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int cb = ks & 1, nb = cb ^ 1;
            if (ks < KS - 1) {
                const int c = 2 * (ks + 1) + half;
#pragma unroll
                for (int i = 0; i < TM; ++i)
                    af[nb][i] = *reinterpret_cast<const bf16x8_t*>(sbase + aoff[i] + ((c ^ aswz[i]) << 4));
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    bfr[nb][j] = *reinterpret_cast<const bf16x8_t*>(sbase + boff[j] + ((c ^ bswz[j]) << 4));
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    if (ROW_ON_LANE)  // D[n][m]: lane = m (row of C), registers = n
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bfr[cb][j], af[cb][i], acc[i][j], 0, 0, 0);
                    else              // D[m][n]: lane = n (column of C), registers = m
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cb][i], bfr[cb][j], acc[i][j], 0, 0, 0);
                }
        }
        cur = (cur + 1 == NS) ? 0 : cur + 1;
    }
}

template <int BM, int BN, int WVM, int WVN, int NS, int BK, int EPI, bool LW = false>
__global__ __launch_bounds__(WVM * WVN * 64 * (LW ? 2 : 1), 2) void gemm_bf16_kernel(GemmArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int WM = BM / WVM, WN = BN / WVN, TM = WM / 32, TN = WN / 32;
    const int tiles_n = p.N / BN;
    // Workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2). Give each XCD
    // a contiguous run of tiles (whole row panels of A, all of W) instead of every 8th tile, so
    // its 4 MiB L2 holds its own slice of A plus W; otherwise every XCD streams all of A and W and
    // the LDS-DMA loads run at the Infinity-Cache rate (~33 GB/s per CU instead of ~70). Speed
    // only: any placement computes the same tile set (bijective for any grid size).
    int bid = blockIdx.x;
    {
        const int nwg = gridDim.x, q8 = nwg >> 3, r8 = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + idx;
    }
    // Which operand an XCD's run shares (round 6): row panels x all columns re-reads ALL of W in every XCD (one stream, fc1:
    // 8 x 4.7 MB for 1.1 MB of A); with few row panels a run of whole COLUMNS reads W / 8 + A per XCD instead (tile_order 2:
    // fc1 of one stream 13.4 -> 11.7 us, QKV 9.6 -> 9.1; the residual GEMMs, A as large as W, keep the row runs; 2 x 4 and
    // 4 x 2 blocks of the tile grid per XCD measured no better: profiles/r06_single_stream_experiments.txt)
    int m0, n0;
    if (p.tile_order == 2) {
        const int tiles_m = (p.M + BM - 1) / BM;
        m0 = (bid % tiles_m) * BM;
        n0 = (bid / tiles_m) * BN;
    } else {
        m0 = (bid / tiles_n) * BM;
        n0 = (bid % tiles_n) * BN;
    }
    if constexpr (LW) {
        if (__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) >= 4) {     // loader waves (see gemm_lw_loader)
            gemm_lw_loader<BM, BN, NS, BK, EPI == EPI_RELU_BF16>(p, smem, m0, n0);
            constexpr bool XEPI = EPI == EPI_F32_POS || EPI == EPI_RESID || EPI == EPI_F32;
            constexpr bool FITS_ = BM * (BN * 2 + 16) <= ring_bytes(NS, BM, BN, BK) && BN * (BM * 2 + 16) <= ring_bytes(NS, BM, BN, BK);
            if constexpr (XEPI || FITS_) { __syncthreads(); __syncthreads(); }    // the epilogue's two workgroup barriers
            // s_endpgm here, not a `return` merged with the computing waves' exit: a merged exit makes hipcc lay the
            // loader code out BEHIND the computing path, guarded by a flag, and the build-time check of the hand-waited
            // loads (tests/test_isa_asm_loads.py, path-insensitive) then sees an infeasible path into it
            __builtin_amdgcn_endpgm();
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave / WVN, wc = wave % WVN, l31 = lane & 31, half = lane >> 5;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    if constexpr (EPI == EPI_F32_POS || EPI == EPI_RESID || EPI == EPI_F32) {
        // ---- X-epilogues: the residual stream as the 3-byte pair (bf16 + lo8, vt_common.hpp) + chunk statistics ----
        // The accumulators (row on the lane, 4 consecutive columns per register quad) are staged as a
        // float32 tile [BM][BN] in the dead operand ring (16-B chunk c of row r at c ^ (r & 7)) and written
        // out row-wise, 8 consecutive columns per lane: addend, bias, statistics of the 32-column chunk a
        // quad of lanes covers, split into hi / lo, two 16-B stores. PIECES 8-column groups per thread.
        constexpr int NT = WVM * WVN * 64, CPR8 = BN / 8, PIECES = BM * BN / 8 / NT;
        static_assert(BM * BN * 4 <= ring_bytes(NS, BM, BN, BK), "f32 tile must fit the operand ring");
        static_assert(NT % CPR8 == 0 && PIECES >= 1, "a thread's column group is the same for all its pieces");
        const int tid = threadIdx.x;
        const int ch8 = tid % CPR8, n8 = n0 + ch8 * 8;
        // 64x64 tiles (small batches: a one-stream GEMM is ~8 us of mostly dependent latencies): the
        // epilogue's addends (old residual pair / positional rows) and the bias do not depend on the
        // product, so they are fetched BEFORE the main loop and their latency runs under it instead of
        // being a round trip at the end of the kernel. Inline-asm global loads: plain loads are sunk to
        // their use behind the loop by hipcc, volatile ones are each followed by vmcnt(0). Being the oldest
        // entries of the in-order vector-memory queue they are retired by the loop's first counted wait;
        // the registers are not touched until the vmcnt(0) below. Rows are clamped: every address is valid.
        constexpr bool EARLY = TM * TN == 1 && EPI != EPI_F32;
        u32x4_t e_a[PIECES][2], e_bias[2];
        u32x2_t e_l[PIECES] = {};          // EPI_RESID: the addend's lo8 bytes (e_a[k][1] is not used then)
        if constexpr (EARLY) {
            e_bias[0] = gload_b128_asm(p.bias + n8);
            e_bias[1] = gload_b128_asm(p.bias + n8 + 4);
#pragma unroll
            for (int k = 0; k < PIECES; ++k) {
                const int m = m0 + (tid + k * NT) / CPR8;
                const int mc = m < p.M ? m : p.M - 1;
                if constexpr (EPI == EPI_F32_POS) {
                    const float* src = p.pos + (size_t)(mc % p.pos_rows) * p.ldx + n8;
                    e_a[k][0] = gload_b128_asm(src);
                    e_a[k][1] = gload_b128_asm(src + 4);
                } else {
                    e_a[k][0] = gload_b128_asm(p.Xh + (size_t)mc * p.ldx + n8);
                    e_l[k] = gload_b64_asm(p.Xl + (size_t)mc * p.ldx + n8);
                }
            }
        }
        if constexpr (LW) gemm_lw_compute<BM, BN, WVM, WVN, NS, BK, true>(p, smem, acc);
        else gemm_mainloop<BM, BN, WVM, WVN, NS, BK, true>(p, smem, m0, n0, acc);
        __syncthreads();                       // every wave has finished reading the ring
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mr = wr * WM + i * 32 + l31;
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int c = (wc * WN + j * 32 + 8 * q + 4 * half) >> 2;      // 16-B chunk of the row
                    *reinterpret_cast<f32x4_t*>(smem + mr * (BN * 4) + ((c ^ (mr & 7)) << 4)) =
                        f32x4_t{acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                }
        }
        __syncthreads();
        float bias8[8];
        if constexpr (EARLY) {
            vm_drain();
            if constexpr (EPI == EPI_RESID) {          // hi (16 B) + lo8 (8 B) per piece
                if constexpr (PIECES == 2)
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_bias[0]), "+v"(e_bias[1]), "+v"(e_a[0][0]), "+v"(e_l[0]),
                                 "+v"(e_a[1][0]), "+v"(e_l[1]) : : "memory");
                else
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_bias[0]), "+v"(e_bias[1]), "+v"(e_a[0][0]), "+v"(e_l[0]) : : "memory");
            } else if constexpr (PIECES == 2)
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_bias[0]), "+v"(e_bias[1]), "+v"(e_a[0][0]), "+v"(e_a[0][1]),
                             "+v"(e_a[1][0]), "+v"(e_a[1][1]) : : "memory");
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(e_bias[0]), "+v"(e_bias[1]), "+v"(e_a[0][0]), "+v"(e_a[0][1]) : : "memory");
            static_assert(!EARLY || PIECES <= 2, "64x64 tile, 256 threads");
            const uint32_t b[8] = {e_bias[0][0], e_bias[0][1], e_bias[0][2], e_bias[0][3], e_bias[1][0], e_bias[1][1], e_bias[1][2], e_bias[1][3]};
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = __uint_as_float(b[e]);
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) bias8[e] = p.bias ? p.bias[n8 + e] : 0.0f;
        }
        const int nchunk = p.N / VT_STAT_CHUNK;
#pragma unroll
        for (int k = 0; k < PIECES; ++k) {
            const int row = (tid + k * NT) / CPR8, m = m0 + row;
            const int mc = m < p.M ? m : p.M - 1;
            const f32x4_t f0 = *reinterpret_cast<const f32x4_t*>(smem + row * (BN * 4) + (((2 * ch8) ^ (row & 7)) << 4));
            const f32x4_t f1 = *reinterpret_cast<const f32x4_t*>(smem + row * (BN * 4) + (((2 * ch8 + 1) ^ (row & 7)) << 4));
            float add[8];
            if constexpr (EPI == EPI_F32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) add[e] = 0.0f;
            } else if constexpr (EPI == EPI_F32_POS) {
                u32x4_t a0, a1;
                if constexpr (EARLY) { a0 = e_a[k][0]; a1 = e_a[k][1]; }
                else {
                    const float* src = p.pos + (size_t)(mc % p.pos_rows) * p.ldx + n8;
                    a0 = *reinterpret_cast<const u32x4_t*>(src); a1 = *reinterpret_cast<const u32x4_t*>(src + 4);
                }
                const uint32_t u[8] = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
#pragma unroll
                for (int e = 0; e < 8; ++e) add[e] = __uint_as_float(u[e]);
            } else {
                u32x4_t hi;
                u32x2_t lo;
                if constexpr (EARLY) { hi = e_a[k][0]; lo = e_l[k]; }
                else {
                    hi = *reinterpret_cast<const u32x4_t*>(p.Xh + (size_t)mc * p.ldx + n8);
                    lo = *reinterpret_cast<const u32x2_t*>(p.Xl + (size_t)mc * p.ldx + n8);
                }
                x_join8(hi, lo, add);
            }
            float x[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = (f0[e] + bias8[e]) + add[e];
                x[4 + e] = (f1[e] + bias8[4 + e]) + add[4 + e];
            }
            float csum, cm2;
            x_chunk_stats(x, csum, cm2);
            u32x4_t hi;
            u32x2_t lo;
            x_split8(x, hi, lo);
            if (m < p.M) {
                *reinterpret_cast<u32x4_t*>(p.Xh + (size_t)m * p.ldx + n8) = hi;
                *reinterpret_cast<u32x2_t*>(p.Xl + (size_t)m * p.ldx + n8) = lo;
                if (p.cstat && (ch8 & 3) == 0)
                    p.cstat[(size_t)m * nchunk + (n8 / VT_STAT_CHUNK)] = make_float2(csum, cm2);
            }
        }
    } else {
        // ---- bf16 outputs -----------------------------------------------------------------------
        // The accumulator layout gives each lane 4 consecutive outputs (8 B) in 32 different rows;
        // stored directly that is one 8-B L2 write request per lane (measured: as many L2 requests
        // as all operand loads of the main loop together). Instead the block's output tile is
        // assembled in LDS (the operand ring is dead by now) and written out as whole 16-B pieces
        // of contiguous rows.
        constexpr int NT = WVM * WVN * 64;
        constexpr bool FITS = BM * (BN * 2 + 16) <= ring_bytes(NS, BM, BN, BK) &&
                              BN * (BM * 2 + 16) <= ring_bytes(NS, BM, BN, BK);
        bool v_tile = false;        // QKV: this column tile holds V (stored transposed)
        float scale = 1.0f;
        if constexpr (EPI == EPI_QKV) {
            v_tile = n0 >= 2 * p.D;
            scale = (n0 < p.D) ? ATT_Q_SCALE : 1.0f;   // q * log2(e)/sqrt(64): scores in log2 units
        }
        // folded LayerNorm: lane l31 (both halves) owns the terms of row l31 of each of its 32-row blocks;
        // what they are made from is fetched before the main loop (k_gemm_util.hpp)
        const bool ln = p.rowstat != nullptr || p.cstat_in != nullptr;
        LnPart lnp[TM];
        if (ln) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + wr * WM + i * 32 + l31;
                ln_prefetch(p, m < p.M ? m : p.M - 1, half, lnp[i]);
            }
        }
        if (!v_tile) {
            // row-major [m][n]: MFMA with the row on the lane, 4 consecutive n per register quad
            if constexpr (LW) gemm_lw_compute<BM, BN, WVM, WVN, NS, BK, true>(p, smem, acc);
            else gemm_mainloop<BM, BN, WVM, WVN, NS, BK, true, EPI == EPI_RELU_BF16>(p, smem, m0, n0, acc);
            constexpr int STRIDE = BN * 2 + 16;
            if constexpr (FITS) __syncthreads();
            // folded LayerNorm: y = a_r * acc + (b_r * colsum[n] + bias[n]); without one a_r = 1, b_r = 0
            // and fma(1, acc, bias) = acc + bias exactly
            // unconditional: under `if (ln)` the waited and the unwaited registers meet in a phi whose copies
            // hipcc places in front of the wait
#pragma unroll
            for (int i = 0; i < TM; ++i) ln_wait(lnp[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int mr = wr * WM + i * 32 + l31;
                float2 rs = make_float2(1.0f, 0.0f);
                if (ln) rs = ln_finish(p, m0 + mr < p.M ? m0 + mr : p.M - 1, lnp[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int nr = wc * WN + j * 32 + 8 * q + 4 * half;
                        float v[4], y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float t = ln ? __builtin_fmaf(rs.y, p.colsum[n0 + nr + e], p.bias[n0 + nr + e])
                                               : p.bias[n0 + nr + e];
                            y[e] = __builtin_fmaf(rs.x, acc[i][j][4 * q + e], t);
                        }
                        if constexpr (EPI == EPI_GELU_BF16) {      // element pairs: packed-f32 polynomial (k_gemm_util.hpp)
#pragma unroll
                            for (int e = 0; e < 4; e += 2) {
                                const f32v2_t g = gelu_erf2(f32v2_t{y[e], y[e + 1]});
                                v[e] = g.x; v[e + 1] = g.y;
                            }
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                if constexpr (EPI == EPI_RELU_BF16) v[e] = fmaxf(y[e], 0.0f);
                                else v[e] = y[e] * scale;
                            }
                        }
                        const uint2 pk = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                        if constexpr (FITS) {
                            *reinterpret_cast<uint2*>(smem + mr * STRIDE + nr * 2) = pk;
                        } else if (m0 + mr < p.M) {
                            bf16_t* dst = (EPI == EPI_QKV)
                                              ? p.qk + (size_t)(m0 + mr) * (2 * p.D) + n0 + nr
                                              : p.Cb + (size_t)(m0 + mr) * p.ldcb + n0 + nr;
                            *reinterpret_cast<uint2*>(dst) = pk;
                        }
                    }
            }
            if constexpr (FITS) {
                __syncthreads();
                constexpr int CH = BN / 8;   // 16-B pieces per row
                for (int c = threadIdx.x; c < BM * CH; c += NT) {
                    const int r = c / CH, ch = c % CH;
                    if (m0 + r < p.M) {
                        const uint4 v = *reinterpret_cast<const uint4*>(smem + r * STRIDE + ch * 16);
                        bf16_t* dst = (EPI == EPI_QKV)
                                          ? p.qk + (size_t)(m0 + r) * (2 * p.D) + n0 + ch * 8
                                          : p.Cb + (size_t)(m0 + r) * p.ldcb + n0 + ch * 8;
                        *reinterpret_cast<uint4*>(dst) = v;
                    }
                }
            }
        } else if constexpr (EPI == EPI_QKV) {
            // V, transposed per head: Vt[b][h][d][t] with t contiguous (npad per row). MFMA with the
            // column (d) on the lane, 4 consecutive tokens per register quad.
            if constexpr (LW) gemm_lw_compute<BM, BN, WVM, WVN, NS, BK, false>(p, smem, acc);
            else gemm_mainloop<BM, BN, WVM, WVN, NS, BK, false>(p, smem, m0, n0, acc);
            constexpr int STRIDE = BM * 2 + 16;
            const int heads = p.D >> 6;
            // folded LayerNorm: the row terms of the 16 rows per 32-row block this lane's registers hold,
            // fetched once (not per column group); rows past M read the last row, never stored
            float2 vrs[TM][4][4];
            // unconditional: under `if (ln)` the waited and the unwaited registers meet in a phi whose copies
            // hipcc places in front of the wait
#pragma unroll
            for (int i = 0; i < TM; ++i) ln_wait(lnp[i]);
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                // lane l31 (both halves) works out the terms of row l31 of the 32-row block, then every lane
                // picks those of the 16 rows its registers hold
                float2 own = make_float2(1.0f, 0.0f);
                if (ln) {
                    const int m = m0 + wr * WM + i * 32 + l31;
                    own = ln_finish(p, m < p.M ? m : p.M - 1, lnp[i]);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int src = 8 * q + 4 * half + e;       // row of the block = lane that owns it
                        vrs[i][q][e] = make_float2(__shfl(own.x, src), __shfl(own.y, src));
                    }
            }
            if constexpr (FITS) __syncthreads();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int nr = wc * WN + j * 32 + l31;
                    const float bias = p.bias[n0 + nr], cs = ln ? p.colsum[n0 + nr] : 0.0f;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int mr = wr * WM + i * 32 + 8 * q + 4 * half;
                        float y[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e)       // rows mr .. mr+3 (tokens), this lane's column d
                            y[e] = __builtin_fmaf(vrs[i][q][e].x, acc[i][j][4 * q + e], __builtin_fmaf(vrs[i][q][e].y, cs, bias));
                        // permuted Vt layout (attention mode 3): the 4-token run moves inside its
                        // group of 16. Streams on 16-token boundaries: permute the tile-local index
                        // here; otherwise (e.g. 980 tokens) each run is placed by its own stream's
                        // token index when it is written out
                        const bool perm_general = p.vt_perm && (p.tokens & 15);
                        const int mp = (p.vt_perm && !perm_general) ? attn_perm16(mr) : mr;
                        const uint2 pk = make_uint2(pack_bf16x2(y[0], y[1]), pack_bf16x2(y[2], y[3]));
                        if constexpr (FITS) {
                            *reinterpret_cast<uint2*>(smem + nr * STRIDE + mp * 2) = pk;
                        } else if (m0 + mr < p.M) {
                            const int m = m0 + mp, nv = n0 + nr - 2 * p.D;
                            const int b = m / p.tokens, t = m % p.tokens;
                            bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad +
                                          (perm_general ? attn_perm16(t) : t);
                            *reinterpret_cast<uint2*>(dst) = pk;
                        }
                    }
                }
            if constexpr (FITS) {
                __syncthreads();
                if (p.vt_perm && (p.tokens & 15)) {      // run by run, 8-B stores (see above)
                    constexpr int CR = BM / 4;           // 4-token runs per d-row
                    for (int c = threadIdx.x; c < BN * CR; c += NT) {
                        const int r = c / CR, cr = c % CR;
                        const int m = m0 + cr * 4, nv = n0 + r - 2 * p.D;
                        if (m >= p.M) continue;
                        const int b = m / p.tokens, t = m % p.tokens;
                        bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + attn_perm16(t);
                        *reinterpret_cast<uint2*>(dst) = *reinterpret_cast<const uint2*>(smem + r * STRIDE + cr * 8);
                    }
                    return;
                }
                constexpr int CH = BM / 8;   // 16-B pieces (8 tokens) per d-row
                for (int c = threadIdx.x; c < BN * CH; c += NT) {
                    const int r = c / CH, ch = c % CH;
                    const int m = m0 + ch * 8, nv = n0 + r - 2 * p.D;
                    if (m >= p.M) continue;
                    const int b = m / p.tokens, t = m % p.tokens;
                    bf16_t* dst = p.vt + ((size_t)(b * heads + (nv >> 6)) * 64 + (nv & 63)) * p.npad + t;
                    const char* src = smem + r * STRIDE + ch * 16;
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
}

// Tile configurations {BM, BN, waves M x N, ring depth, K-tile depth} of this file's kernel:
//   0: 64x64   2x2 ring 4 k64    small M, long K (fc2 of two or three streams)
//   1: 128x128 2x2 ring 3 k64
//   2: 64x64   2x2 ring 2 k64    small M: most blocks per CU
//   3: 128x128 2x2 ring 2 k64
//   4: 64x64   2x2 ring 3 k128   one stream: proj, fc2 (half the iterations of the dependent K loop)
//   5: 64x64   2x2 ring 2 k128   one stream: QKV; two or three streams: proj
//   6: 128x128 2x2 ring 2 k128   (measured, never the best: kept for the sweeps)
// 18 / 19 are the 256x256 8-wave kernels of k_gemm256.hip (17: round 1's schedule, removed). Numbers 4-16 were experiments on this kernel
// (256-wide tiles with the same loop, 8-wave blocks, deeper rings, K-tile depth 32, register
// staging) that never beat 128x128 ring 2 and were removed after the sweep kept in
// profiles/gemm_sweep_r01.txt.
//   7: 64x64   ring 3 k128 + 4 LOADER waves (round 5: 512 threads, gemm_lw_loader / gemm_lw_compute): the one-stream
//              residual GEMMs (proj, fc2: 144 tiles, one long tile per CU)
//   8: 128x64  ring 3 k128 + 4 loader waves: fc2 of two / three streams
// (measured and dropped, profiles/r05_loader_wave_small_batch.txt: 64x64 ring 4 / 6 k64 and ring 4 k128, 128x64 ring 4 k64,
//  128x128 ring 4 k64 with loader waves; round 6, profiles/r06_single_stream_experiments.txt: 96x96 tiles on three waves -
//  fc1 of one stream as ONE round of 256 tiles - ring 4 k64 and ring 3 k128: 15.2 us against 13.2)
#define GEMM_FOR_EACH_CFG(X, EPI) \
    X(0, 64, 64, 2, 2, 4, 64, EPI, false)    \
    X(1, 128, 128, 2, 2, 3, 64, EPI, false)  \
    X(2, 64, 64, 2, 2, 2, 64, EPI, false)    \
    X(3, 128, 128, 2, 2, 2, 64, EPI, false)  \
    X(4, 64, 64, 2, 2, 3, 128, EPI, false)   \
    X(5, 64, 64, 2, 2, 2, 128, EPI, false)   \
    X(6, 128, 128, 2, 2, 2, 128, EPI, false) \
    X(7, 64, 64, 2, 2, 3, 128, EPI, true)    \
    X(8, 128, 64, 2, 2, 3, 128, EPI, true)
#define GEMM_NUM_CFG 20   // valid: 0..8 (this file) and 18, 19 (k_gemm256.hip)

template <int BM, int BN, int WVM, int WVN, int NS, int BK, int EPI, bool LW>
static hipError_t prepare_cfg() {
    constexpr int smem = ring_bytes(NS, BM, BN, BK);
    return hipFuncSetAttribute(
        reinterpret_cast<const void*>(&gemm_bf16_kernel<BM, BN, WVM, WVN, NS, BK, EPI, LW>),
        hipFuncAttributeMaxDynamicSharedMemorySize, smem);
}

template <int EPI>
static hipError_t prepare_epi() {
    hipError_t e = hipSuccess;
#define X(id, BM, BN, WVM, WVN, NS, BK, E, LW) \
    if (e == hipSuccess) e = prepare_cfg<BM, BN, WVM, WVN, NS, BK, E, LW>();
    GEMM_FOR_EACH_CFG(X, EPI)
#undef X
    return e;
}

// Raise the dynamic-LDS limit of every instantiation once per device, outside any stream capture.
hipError_t gemm_prepare() {
    hipError_t e;
    if ((e = prepare_epi<EPI_F32_POS>()) != hipSuccess) return e;
    if ((e = prepare_epi<EPI_RESID>()) != hipSuccess) return e;
    if ((e = prepare_epi<EPI_GELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_epi<EPI_RELU_BF16>()) != hipSuccess) return e;
    if ((e = prepare_epi<EPI_QKV>()) != hipSuccess) return e;
    if ((e = prepare_epi<EPI_F32>()) != hipSuccess) return e;
    return gemm256_prepare();
}

template <int BM, int BN, int WVM, int WVN, int NS, int BK, int EPI, bool LW>
static hipError_t launch_cfg(const GemmArgs& a, hipStream_t st) {
    constexpr int smem = ring_bytes(NS, BM, BN, BK);
    if (a.N % BN != 0 || a.K % BK != 0) return hipErrorInvalidValue;
    if (a.conv_grid > 0 && a.conv_C % BK != 0) return hipErrorInvalidValue;   // a K-tile lies inside one tap
    if (EPI == EPI_QKV && a.D % BN != 0) return hipErrorInvalidValue;  // a column tile is q, k or v
    const int tiles = ((a.M + BM - 1) / BM) * (a.N / BN);
    GemmArgs b = a;
    if (b.tile_order < 1 || b.tile_order > 2)     // QKV / fc1 of a few streams: A is the smaller operand, an XCD shares it whole
        b.tile_order = ((EPI == EPI_QKV || EPI == EPI_GELU_BF16) && a.M < a.N) ? 2 : 1;
    vt_launch((gemm_bf16_kernel<BM, BN, WVM, WVN, NS, BK, EPI, LW>), dim3(tiles),
                       dim3(WVM * WVN * 64 * (LW ? 2 : 1)), smem, st, b);
    return hipGetLastError();
}

// Chosen from sweeps on MI355X over the tracker's shapes (M = 720 * streams; profiles/
// gemm_sweep_r01.txt).
int gemm_pick_config(int M, int N, int K, int epilogue, bool conv) {
    const long tiles128 = (long)((M + 127) / 128) * (N / 128);
    const bool n128 = (N % 128) == 0;
    // 256x256 8-wave kernel (k_gemm256.hip): one workgroup per CU, so it wants the grid to fill
    // the 256 CUs in whole rounds; measured against 128x128 on the tracker's shapes it wins from
    // about half a round upwards, also when the last round is nearly empty (fc2 at 31 streams, 264
    // tiles = 2 rounds: 133 us against ~160; profiles/gemm_sweep_r01.txt). 19 = schedule v2 with
    // persistent workgroups where that applies (bf16 outputs, more tiles than CUs), else the
    // one-tile-per-workgroup launch of the same schedule (= 18). Implicit convolutions (the head's
    // 3x3 layers) gather their A rows in the 4-wave kernel only.
    if (!conv && (N % 256) == 0 && K >= 128) {
        const long t = (long)((M + 255) / 256) * (N / 256);
        if (t >= 128) return GEMM_CFG_256PP;
    }
    // Below that the 4-wave kernel (round-2 sweep at 1-8 streams, profiles/r02_small_batch_gemm_ab.txt):
    // up to 128 tiles of 128x128 the 64x64 tile with the deepest ring wins (most workgroups, three
    // K-tiles in flight); up to 256 every 128x128 tile has a CU to itself and the ring of 3 (96 KB, one
    // workgroup per CU) beats the ring of 2; beyond, two workgroups per CU need the ring of 2.
    // One or two streams are launch- and latency-bound (a K = 768 GEMM of 60 workgroups still takes 7 us):
    // there a K-tile depth of 128 (configs 4, 5: half the barriers and waits of the dependent K loop)
    // is worth 0.5 us on the K = 768 shapes and 2-3 us on fc2.
    const bool k128 = (K % 128) == 0;
    switch (epilogue) {
        // (re-measured in round 3 with the folded-LayerNorm epilogues, profiles/r03_small_batch_gemm_sweep.txt:
        // one stream QKV 5: 9.5 us, fc1 2: 13.4 (1: 16.6); two streams QKV 2: 14.8 (1: 16.6), fc1 3: 22.3)
        case EPI_QKV:
            if (!n128) return 2;
            return tiles128 <= 128 ? (k128 ? 5 : 0) : (tiles128 <= 256 ? 2 : 3);
        case EPI_GELU_BF16:
            if (!n128) return 2;
            return tiles128 <= 128 ? 0 : (tiles128 <= 192 ? 2 : (tiles128 <= 256 ? 1 : 3));
        case EPI_RELU_BF16:             // head convs: N = 128
            if (n128 && tiles128 >= 256) return 3;
            // a few streams: 9-18 workgroups walk 12-18 K-tiles one after the other - a dependent chain of
            // L2 round trips; the deepest prefetch (K-tile depth 128, ring of 3) halves it
            if (tiles128 <= 64 && k128 && (!conv || (N % 128) == 0)) return 4;
            return 2;
        case EPI_RESID:
            // round 5 (profiles/r05_loader_wave_small_batch.txt): with four loader waves beside the four computing
            // waves (configs 7, 8) the one-stream residual GEMMs - 144 tiles, one per CU, each a long K loop - run at
            // the CU's LDS fill rate: fc2 14.5 -> 12.7 us, proj 6.1 -> 5.5 (config 7); fc2 of two / three streams
            // 21.3 -> 19.8 / 22.4 -> 20.9 (config 8). QKV and fc1 (2-3 short tiles per CU) do not gain and keep theirs.
            if (K >= 2048) {                       // fc2: long dependent K loop
                if (M <= 1024 && k128) return 7;
                if (M <= 2304 && k128) return 8;
                if (M <= 2304) return 0;
                if (n128) return tiles128 < 256 ? 1 : 3;
            }
            if (k128 && M <= 1024) return 7;       // proj, one stream
            if (k128 && M <= 2304) return 5;
            return (n128 && tiles128 >= 256) ? 3 : 2;     // proj: 8-14 streams (beyond, the 256x256 kernel)
        default: return 2;
    }
}

const char* gemm_config_name(int cfg) {
    static const char* n[] = {"64x64x4", "128x128x3", "64x64x2", "128x128x2", "64x64x3k128", "64x64x2k128", "128x128x2k128",
                              "64x64x3k128lw", "128x64x3k128lw"};
    if (cfg == GEMM_CFG_256P4) return "256x256p4";
    if (cfg == GEMM_CFG_256PP) return "256x256pp";
    return (cfg >= 0 && cfg <= GEMM_CFG_SMALL_MAX) ? n[cfg] : "?";
}

template <int EPI>
static hipError_t launch_epi(const GemmArgs& a, int cfg, hipStream_t st) {
    switch (cfg) {
#define X(id, BM, BN, WVM, WVN, NS, BK, E, LW) \
    case id: return launch_cfg<BM, BN, WVM, WVN, NS, BK, E, LW>(a, st);
        GEMM_FOR_EACH_CFG(X, EPI)
#undef X
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_gemm_cfg(const GemmArgs& a, int epilogue, int cfg, hipStream_t st) {
    if (a.M <= 0 || a.N % 64 != 0 || a.K % GEMM_BK != 0 || a.K <= 0) return hipErrorInvalidValue;
    if (a.conv_grid > 0 && (epilogue != EPI_RELU_BF16 || cfg > GEMM_CFG_SMALL_MAX || a.conv_C % GEMM_BK != 0 || a.K != 9 * a.conv_C ||
                            a.lda != a.conv_C || !a.zeros || a.M % (a.conv_grid * a.conv_grid) != 0))
        return hipErrorInvalidValue;
    if (epilogue == EPI_QKV && (a.D % 64 != 0 || (a.tokens & 3) != 0 || (a.npad & 3) != 0))
        return hipErrorInvalidValue;
    const bool x_epi = epilogue == EPI_F32_POS || epilogue == EPI_RESID || epilogue == EPI_F32;
    if (x_epi && (!a.Xh || !a.Xl || (a.ldx & 7) || (epilogue == EPI_F32_POS && (!a.pos || a.pos_rows < 1))))
        return hipErrorInvalidValue;
    if (!x_epi && (!a.bias || ((a.rowstat || a.cstat_in) && !a.colsum))) return hipErrorInvalidValue;
    if (a.cstat_in && (cfg > GEMM_CFG_SMALL_MAX || a.rowstat || (a.K % (4 * VT_STAT_CHUNK)) != 0 || a.K > 1024)) return hipErrorInvalidValue;
    if (cfg == GEMM_CFG_256P4) return launch_gemm256(a, epilogue, false, st);
    if (cfg == GEMM_CFG_256PP) return launch_gemm256(a, epilogue, true, st);
    switch (epilogue) {
        case EPI_F32_POS: return launch_epi<EPI_F32_POS>(a, cfg, st);
        case EPI_RESID: return launch_epi<EPI_RESID>(a, cfg, st);
        case EPI_GELU_BF16: return launch_epi<EPI_GELU_BF16>(a, cfg, st);
        case EPI_RELU_BF16: return launch_epi<EPI_RELU_BF16>(a, cfg, st);
        case EPI_QKV: return launch_epi<EPI_QKV>(a, cfg, st);
        case EPI_F32: return launch_epi<EPI_F32>(a, cfg, st);
        default: return hipErrorInvalidValue;
    }
}

int gemm_effective_config(const GemmArgs& a, int epilogue) {
    const int cfg = gemm_pick_config(a.M, a.N, a.K, epilogue, a.conv_grid > 0);
    if (cfg >= GEMM_CFG_256_MIN && !gemm256_fits(a, epilogue))       // e.g. an operand beyond 4 GiB
        return (a.N % 128 == 0 && a.M >= 2048) ? 3 : 2;
    return cfg;
}

bool gemm_finalizes_rowstat(const GemmArgs& a, int epilogue) {
    const bool x_epi = epilogue == EPI_F32_POS || epilogue == EPI_RESID || epilogue == EPI_F32;
    return x_epi && a.rowstat_out && a.panel_cnt && a.cstat && a.N <= 1536 && gemm_effective_config(a, epilogue) >= GEMM_CFG_256_MIN;
}

hipError_t launch_gemm(const GemmArgs& a, int epilogue, hipStream_t st) {
    const int cfg = gemm_effective_config(a, epilogue);
    hipError_t e = launch_gemm_cfg(a, epilogue, cfg, st);
    if (e == hipErrorInvalidValue && cfg != 2) e = launch_gemm_cfg(a, epilogue, 2, st);  // shape does not fit the chosen tile
    return e;
}
