// vt_ops.hip — operator-level entry points (include/vittrack_hip_ops.h): numerics tests and tuning tools call the
// same kernel launchers the pass uses. Linked into libvittrack_hip_ops.so only; the product library does not carry them.
#include "vt_engine.hpp"
#include "../../include/vittrack_hip_ops.h"

extern "C" {

// ---- operator-level entry points ---------------------------------------------------------------------------

// host-side helpers of the operator entry points: float32 <-> the 3-byte pair of the residual stream
static inline bf16_t host_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    return (bf16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static inline float host_f32(bf16_t b) {
    const uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

// epilogue: 0 x = acc + bias, 1 x = (acc + bias) + c_inout, 4 x = (acc + bias) + pos (pos = c_inout, one
// row per output row) - the X-epilogues: c_inout goes in and comes back through the 3-byte pair (hi + lo8 * 2^-12: bf16 and
// an absolute quantum of 2^-12), rowstat_out (if given) receives the finalized row terms (rstd, -mean * rstd) of x;
// 2 gelu, 3 relu -> bf16, with an optional folded LayerNorm (rowstat_in [M][2], colsum [N]).
int vt_op_gemm_bf16(int device_id, const uint16_t* a, const uint16_t* w, const float* bias, float* c_inout,
                    int M, int N, int K, int epilogue, int cfg, const float* rowstat_in, const float* colsum,
                    float* rowstat_out, float eps) try {
    if (!a || !w || !c_inout || M <= 0 || N <= 0 || K <= 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (N % 64 || K % 64) return set_err(VT_ERR_INVALID_ARG, "gemm: N and K must be multiples of 64");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(gemm_prepare()); HIPCHK(attention_prepare()); HIPCHK(headconv_prepare());
    const size_t MN = (size_t)M * N;
    DevBuf da, dw, db, dxh, dxl, dpos, dcb, dcs, drs, dcst, dro;
    HIPCHK(da.alloc((size_t)M * K * 2)); HIPCHK(dw.alloc((size_t)N * K * 2)); HIPCHK(db.alloc((size_t)N * 4));
    HIPCHK(dxh.alloc(MN * 2)); HIPCHK(dxl.alloc(MN * 2)); HIPCHK(dcb.alloc(MN * 2));
    HIPCHK(hipMemcpy(da.p, a, (size_t)M * K * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw.p, w, (size_t)N * K * 2, hipMemcpyHostToDevice));
    std::vector<float> zb((size_t)N, 0.0f);
    HIPCHK(hipMemcpy(db.p, bias ? bias : zb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
    GemmArgs g{};
    g.A = (const bf16_t*)da.p; g.lda = K; g.W = (const bf16_t*)dw.p; g.ldw = K; g.bias = (const float*)db.p;
    g.M = M; g.N = N; g.K = K; g.Xh = (bf16_t*)dxh.p; g.Xl = (uint8_t*)dxl.p; g.ldx = N; g.Cb = (bf16_t*)dcb.p; g.ldcb = N;
    int epi;
    switch (epilogue) {
        case 0: epi = EPI_F32; break;
        case 1: epi = EPI_RESID; break;
        case 2: epi = EPI_GELU_BF16; break;
        case 3: epi = EPI_RELU_BF16; break;
        case 4: epi = EPI_F32_POS; break;
        default: return set_err(VT_ERR_INVALID_ARG, "gemm: unknown epilogue %d", epilogue);
    }
    const bool x_epi = epi == EPI_F32 || epi == EPI_RESID || epi == EPI_F32_POS;
    // the 3-byte pair of specification v3 (vt_common.hpp): hi = bf16(x), lo8 = clamp(rint((x - hi) * 2^12), -127, 127)
    std::vector<bf16_t> hi;
    std::vector<int8_t> lo;
    if (epi == EPI_RESID) {
        hi.resize(MN); lo.resize(MN);
        for (size_t i = 0; i < MN; ++i) {
            hi[i] = host_bf16(c_inout[i]);
            const float q = nearbyintf((c_inout[i] - host_f32(hi[i])) * 4096.0f);
            lo[i] = (int8_t)(q < -127.0f ? -127.0f : q > 127.0f ? 127.0f : q);
        }
        HIPCHK(hipMemcpy(dxh.p, hi.data(), MN * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dxl.p, lo.data(), MN, hipMemcpyHostToDevice));
    } else if (epi == EPI_F32_POS) {
        HIPCHK(dpos.alloc(MN * 4));
        HIPCHK(hipMemcpy(dpos.p, c_inout, MN * 4, hipMemcpyHostToDevice));
        g.pos = (const float*)dpos.p; g.pos_rows = M;
    }
    if (x_epi) {
        HIPCHK(dcst.alloc((size_t)M * (N / VT_STAT_CHUNK) * 8));
        g.cstat = (float2*)dcst.p;
    } else if (rowstat_in) {
        if (!colsum) return set_err(VT_ERR_INVALID_ARG, "gemm: rowstat without colsum");
        HIPCHK(drs.alloc((size_t)M * 8 + 16)); HIPCHK(dcs.alloc((size_t)N * 4));
        HIPCHK(hipMemcpy(drs.p, rowstat_in, (size_t)M * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dcs.p, colsum, (size_t)N * 4, hipMemcpyHostToDevice));
        g.rowstat = (const float2*)drs.p; g.colsum = (const float*)dcs.p;
    }
    // X-epilogues on the 256x256 kernel finalize the row terms themselves (last workgroup of each row panel)
    DevBuf dcnt;
    bool fused = false;
    if (cfg >= 0) { g.tile_order = (cfg >> 8) & 3; cfg &= 0xff; }   // bits 8-9 of a given configuration: the XCD tile order (1 rows, 2 columns)
    if (x_epi && rowstat_out) {
        HIPCHK(dro.alloc((size_t)M * 8));
        HIPCHK(dcnt.alloc((size_t)((M + 255) / 256 + 1) * 4));
        HIPCHK(hipMemset(dcnt.p, 0, (size_t)((M + 255) / 256 + 1) * 4));
        HIPCHK(hipMemset(dro.p, 0xff, (size_t)M * 8));
        const int eff = cfg < 0 ? gemm_effective_config(g, epi) : cfg;
        if (eff >= GEMM_CFG_256_MIN) {
            g.rowstat_out = (float2*)dro.p; g.panel_cnt = (unsigned*)dcnt.p; g.ln_eps = eps;
            fused = true;
        }
    }
    if (cfg < 0) HIPCHK(launch_gemm(g, epi, nullptr));
    else if (launch_gemm_cfg(g, epi, cfg, nullptr) != hipSuccess)
        return set_err(VT_ERR_INVALID_ARG, "gemm: tile configuration %d does not fit M=%d N=%d K=%d", cfg, M, N, K);
    if (x_epi && rowstat_out && !fused)
        HIPCHK(launch_rowstat_finalize(g.cstat, (float2*)dro.p, M, N / VT_STAT_CHUNK, eps, nullptr));
    if (fused) {      // launch it twice more: the counters must come back to zero by themselves
        for (int rep = 0; rep < 2 && (epi == EPI_F32 || epi == EPI_F32_POS); ++rep) HIPCHK(launch_gemm_cfg(g, epi, cfg < 0 ? gemm_effective_config(g, epi) : cfg, nullptr));
    }
    HIPCHK(hipDeviceSynchronize());
    if (x_epi) {
        hi.resize(MN); lo.resize(MN);
        HIPCHK(hipMemcpy(hi.data(), dxh.p, MN * 2, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(lo.data(), dxl.p, MN, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < MN; ++i) c_inout[i] = host_f32(hi[i]) + (float)lo[i] * VT_LO_Q;
        if (rowstat_out) HIPCHK(hipMemcpy(rowstat_out, dro.p, (size_t)M * 8, hipMemcpyDeviceToHost));
    } else {
        std::vector<bf16_t> tmp(MN);
        HIPCHK(hipMemcpy(tmp.data(), dcb.p, tmp.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) c_inout[i] = host_f32(tmp[i]);
    }
    return VT_OK;
} VT_NOTHROW_INT

// Timing helper for kernel tuning: runs the GEMM kernel `iters` times on device-resident random
// operands with tile configuration `cfg` (<0: the launcher's own choice) and returns the mean time
// per launch in microseconds (HIP events on the null stream).
int vt_op_gemm_bench(int device_id, int M, int N, int K, int epilogue, int cfg, int iters, float* us_out) try {
    if (M <= 0 || N % 64 || K % 64 || iters < 1 || !us_out) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(gemm_prepare()); HIPCHK(attention_prepare()); HIPCHK(headconv_prepare());
    const int D = N / 3, tokens = 4 * ((M + 3) / 4), npad = (tokens + 63) / 64 * 64;
    DevBuf da, dw, db, dc, dcb, dvt, dxl, dcst, drs;
    HIPCHK(da.alloc((size_t)M * K * 2)); HIPCHK(dw.alloc((size_t)N * K * 2)); HIPCHK(db.alloc((size_t)N * 4));
    HIPCHK(dc.alloc((size_t)M * N * 4)); HIPCHK(dcb.alloc((size_t)M * N * 2)); HIPCHK(dxl.alloc((size_t)M * N * 2));
    HIPCHK(dcst.alloc((size_t)M * (N / VT_STAT_CHUNK) * 8)); HIPCHK(drs.alloc((size_t)M * 8 + 16));
    HIPCHK(dvt.alloc((size_t)(N / 64 + 1) * 64 * npad * 2));
    std::vector<bf16_t> ha((size_t)M * K), hw((size_t)N * K);
    uint32_t seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (bf16_t)(0x3c00u + ((seed >> 9) & 0x3ffu) + ((seed >> 3) & 0x8000u)); };
    for (auto& v : ha) v = rnd();
    for (auto& v : hw) v = rnd();
    HIPCHK(hipMemcpy(da.p, ha.data(), ha.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw.p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(db.p, 0, (size_t)N * 4));
    HIPCHK(hipMemset(dc.p, 0, (size_t)M * N * 4));
    HIPCHK(hipMemset(dcb.p, 0, (size_t)M * N * 2)); HIPCHK(hipMemset(dxl.p, 0, (size_t)M * N * 2));
    {   // folded-LayerNorm row terms as the engine passes them to the QKV / fc1 GEMMs: (1, 0) per row
        std::vector<float> rs((size_t)M * 2);
        for (int i = 0; i < M; ++i) { rs[2 * (size_t)i] = 1.0f; rs[2 * (size_t)i + 1] = 0.0f; }
        HIPCHK(hipMemcpy(drs.p, rs.data(), rs.size() * 4, hipMemcpyHostToDevice));
    }
    GemmArgs g{};
    g.A = (const bf16_t*)da.p; g.lda = K; g.W = (const bf16_t*)dw.p; g.ldw = K; g.bias = (const float*)db.p;
    g.M = M; g.N = N; g.K = K; g.Cb = (bf16_t*)dcb.p; g.ldcb = N;
    const bool x_epi = epilogue == EPI_F32 || epilogue == EPI_RESID || epilogue == EPI_F32_POS;
    DevBuf dcnt, dro;
    if (x_epi) {      // as the engine launches it: chunk partials + the row terms finalized by the last workgroup of each panel
        g.Xh = (bf16_t*)dcb.p; g.Xl = (uint8_t*)dxl.p; g.ldx = N; g.cstat = (float2*)dcst.p;
        HIPCHK(dcnt.alloc((size_t)((M + 255) / 256 + 1) * 4)); HIPCHK(dro.alloc((size_t)M * 8 + 16));
        HIPCHK(hipMemset(dcnt.p, 0, (size_t)((M + 255) / 256 + 1) * 4));
        g.rowstat_out = (float2*)dro.p; g.panel_cnt = (unsigned*)dcnt.p; g.ln_eps = 1e-6f;
    }
    else if (epilogue == EPI_QKV || epilogue == EPI_GELU_BF16) { g.rowstat = (const float2*)drs.p; g.colsum = (const float*)db.p; }
    g.pos = (const float*)dc.p; g.pos_rows = M;
    g.qk = (bf16_t*)dcb.p; g.vt = (bf16_t*)dvt.p; g.tokens = tokens; g.npad = npad; g.D = D;
    if (epilogue == EPI_QKV && (N % 192 || tokens != M)) return set_err(VT_ERR_INVALID_ARG, "qkv bench: N = 3D, D %% 64 == 0, M %% 4 == 0");
    if (cfg >= 0) { g.tile_order = (cfg >> 8) & 3; cfg &= 0xff; }   // sweeps: bits 8-9 force the XCD tile order (1 rows, 2 columns)
    if (cfg < 0) cfg = gemm_pick_config(M, N, K, epilogue);
    DevBuf ddbg;
    const size_t dbg_words = (size_t)((M + 63) / 64) * (N / 64) * 8 * 4;
#ifdef VT_STAMPS   // diagnostic builds only: per-wave cycle sums of the main loop
    HIPCHK(ddbg.alloc(dbg_words * 8));
    HIPCHK(hipMemset(ddbg.p, 0, dbg_words * 8));
    g.dbg = (unsigned long long*)ddbg.p;
#else
    (void)dbg_words;
#endif
    for (int i = 0; i < 3; ++i) HIPCHK(launch_gemm_cfg(g, epilogue, cfg, nullptr));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) HIPCHK(launch_gemm_cfg(g, epilogue, cfg, nullptr));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_out = ms * 1000.0f / iters;
    if (g.dbg) {   // diagnostic build: mean per-wave cycle split of the main loop
        std::vector<unsigned long long> h(dbg_words);
        HIPCHK(hipMemcpy(h.data(), ddbg.p, dbg_words * 8, hipMemcpyDeviceToHost));
        double s[4] = {0, 0, 0, 0};
        size_t n = 0;
        for (size_t i = 0; i + 3 < dbg_words; i += 4)
            if (h[i + 3]) { for (int k = 0; k < 4; ++k) s[k] += (double)h[i + k]; ++n; }
        if (n) fprintf(stderr, "stamps cfg %d: waves %zu  wait %.0f  issue %.0f  compute %.0f  loop total %.0f cycles/wave\n",
                       cfg, n, s[0] / n, s[1] / n, s[2] / n, s[3] / n);
    }
    return VT_OK;
} VT_NOTHROW_INT

int vt_op_qkv_bf16(int device_id, const uint16_t* a, const uint16_t* w, const float* bias, float* qk_out,
                   float* vt_out, int B, int tokens, int D, int cfg, int vt_perm, const float* rowstat_in,
                   const float* colsum) try {
    // QKV GEMM with the attention-layout epilogue: qk_out [B*tokens][2D], vt_out [B*H][64][npad]
    if (!a || !w || !bias || !qk_out || !vt_out || B <= 0 || tokens <= 0 || D % 64 || (tokens & 3))
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(gemm_prepare()); HIPCHK(attention_prepare()); HIPCHK(headconv_prepare());
    const int M = B * tokens, H = D / 64, npad = (tokens + 63) / 64 * 64;
    DevBuf da, dw, db, dqk, dvt;
    HIPCHK(da.alloc((size_t)M * D * 2)); HIPCHK(dw.alloc((size_t)3 * D * D * 2)); HIPCHK(db.alloc((size_t)3 * D * 4));
    HIPCHK(dqk.alloc((size_t)M * 2 * D * 2)); HIPCHK(dvt.alloc((size_t)B * H * 64 * npad * 2));
    HIPCHK(hipMemcpy(da.p, a, (size_t)M * D * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw.p, w, (size_t)3 * D * D * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, bias, (size_t)3 * D * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dvt.p, 0, (size_t)B * H * 64 * npad * 2));
    GemmArgs g{};
    g.A = (const bf16_t*)da.p; g.lda = D; g.W = (const bf16_t*)dw.p; g.ldw = D; g.bias = (const float*)db.p;
    g.M = M; g.N = 3 * D; g.K = D; g.qk = (bf16_t*)dqk.p; g.vt = (bf16_t*)dvt.p; g.tokens = tokens; g.npad = npad; g.D = D;
    g.vt_perm = vt_perm ? 1 : 0;   // 1: the key order attention mode 3 reads
    DevBuf drs, dcs;
    if (rowstat_in) {              // folded LayerNorm: [M][2] row terms, [3D] column sums
        if (!colsum) return set_err(VT_ERR_INVALID_ARG, "qkv: rowstat without colsum");
        HIPCHK(drs.alloc((size_t)M * 8 + 16)); HIPCHK(dcs.alloc((size_t)3 * D * 4));
        HIPCHK(hipMemcpy(drs.p, rowstat_in, (size_t)M * 8, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dcs.p, colsum, (size_t)3 * D * 4, hipMemcpyHostToDevice));
        g.rowstat = (const float2*)drs.p; g.colsum = (const float*)dcs.p;
    }
    if (cfg >= 0) { g.tile_order = (cfg >> 8) & 3; cfg &= 0xff; }   // as in vt_op_gemm_bf16
    if (cfg < 0) HIPCHK(launch_gemm(g, EPI_QKV, nullptr));
    else if (launch_gemm_cfg(g, EPI_QKV, cfg, nullptr) != hipSuccess)
        return set_err(VT_ERR_INVALID_ARG, "qkv: tile configuration %d does not fit this shape", cfg);
    HIPCHK(hipDeviceSynchronize());
    auto widen = [](const DevBuf& d, size_t count, float* out) -> hipError_t {
        std::vector<bf16_t> tmp(count);
        hipError_t e = hipMemcpy(tmp.data(), d.p, count * 2, hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        for (size_t i = 0; i < count; ++i) { uint32_t u = ((uint32_t)tmp[i]) << 16; memcpy(out + i, &u, 4); }
        return hipSuccess;
    };
    HIPCHK(widen(dqk, (size_t)M * 2 * D, qk_out));
    HIPCHK(widen(dvt, (size_t)B * H * 64 * npad, vt_out));
    return VT_OK;
} VT_NOTHROW_INT

int vt_op_attention_bf16(int device_id, const uint16_t* q, const uint16_t* k, const uint16_t* v, float* out,
                         int B, int N, int H, int mode) try {
    if (!q || !k || !v || !out || B <= 0 || N <= 0 || H <= 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(attention_prepare());
    const int D = H * 64, M = B * N, npad = (N + 63) / 64 * 64;
    // host-side packing into the layouts the QKV epilogue produces
    if (mode < 0) mode = attention_pick_mode(N, npad);
    const bool perm = attention_vt_perm(mode) != 0;
    std::vector<bf16_t> qk((size_t)M * 2 * D), vt((size_t)B * H * 64 * npad, 0);
    for (int m = 0; m < M; ++m) {
        memcpy(&qk[(size_t)m * 2 * D], q + (size_t)m * D, (size_t)D * 2);
        memcpy(&qk[(size_t)m * 2 * D + D], k + (size_t)m * D, (size_t)D * 2);
        const int b = m / N, t = m % N, tp = perm ? attn_perm16(t) : t;
        for (int c = 0; c < D; ++c)
            vt[((size_t)(b * H + c / 64) * 64 + c % 64) * npad + tp] = v[(size_t)m * D + c];
    }
    DevBuf dqk, dvt, dout;
    HIPCHK(dqk.alloc(qk.size() * 2)); HIPCHK(dvt.alloc(vt.size() * 2)); HIPCHK(dout.alloc((size_t)M * D * 2));
    HIPCHK(hipMemcpy(dqk.p, qk.data(), qk.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dvt.p, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(launch_attention_mode((const bf16_t*)dqk.p, (const bf16_t*)dvt.p, (bf16_t*)dout.p, B, N, H, npad, mode, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<bf16_t> tmp((size_t)M * D);
    HIPCHK(hipMemcpy(tmp.data(), dout.p, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) { uint32_t u = ((uint32_t)tmp[i]) << 16; memcpy(out + i, &u, 4); }
    return VT_OK;
} VT_NOTHROW_INT

// Timing helper: mean microseconds per launch of the attention kernel (mode as VT_ATTN_MODE, <0 =
// the launcher's choice) on device-resident random data.
int vt_op_attention_bench(int device_id, int B, int N, int H, int mode, int iters, float* us_out) try {
    if (B <= 0 || N <= 0 || H <= 0 || iters < 1 || !us_out) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(attention_prepare());
    const int D = H * 64, M = B * N, npad = (N + 63) / 64 * 64;
    std::vector<bf16_t> qk((size_t)M * 2 * D), vt((size_t)B * H * 64 * npad);
    uint32_t seed = 777u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (bf16_t)(0x3c00u + ((seed >> 9) & 0x3ffu) + ((seed >> 3) & 0x8000u)); };
    for (auto& v : qk) v = rnd();
    for (auto& v : vt) v = rnd();
    DevBuf dqk, dvt, dout;
    HIPCHK(dqk.alloc(qk.size() * 2)); HIPCHK(dvt.alloc(vt.size() * 2)); HIPCHK(dout.alloc((size_t)M * D * 2));
    HIPCHK(hipMemcpy(dqk.p, qk.data(), qk.size() * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dvt.p, vt.data(), vt.size() * 2, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i)
        HIPCHK(launch_attention_mode((const bf16_t*)dqk.p, (const bf16_t*)dvt.p, (bf16_t*)dout.p, B, N, H, npad, mode, nullptr));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i)
        HIPCHK(launch_attention_mode((const bf16_t*)dqk.p, (const bf16_t*)dvt.p, (bf16_t*)dout.p, B, N, H, npad, mode, nullptr));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_out = ms * 1000.0f / iters;
    return VT_OK;
} VT_NOTHROW_INT

int vt_op_nv12_to_rgb8_bench(int device_id, int w, int h, int iters, float* us_out) try {
    if (w < 2 || h < 2 || w > 16384 || h > 16384 || iters < 1 || !us_out) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    const size_t in_bytes = nv12_bytes_read(w, h) + 16, out_bytes = (size_t)w * h * 3;
    DevBuf din, dout;
    HIPCHK(din.alloc(in_bytes)); HIPCHK(dout.alloc(out_bytes));
    std::vector<uint8_t> host(in_bytes);
    uint32_t seed = 2463534242u;
    for (auto& v : host) { seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5; v = (uint8_t)seed; }
    HIPCHK(hipMemcpy(din.p, host.data(), in_bytes, hipMemcpyHostToDevice));
    for (int i = 0; i < 3; ++i) HIPCHK(launch_nv12_to_rgb8((const uint8_t*)din.p, w, h, (uint8_t*)dout.p, nullptr));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) HIPCHK(launch_nv12_to_rgb8((const uint8_t*)din.p, w, h, (uint8_t*)dout.p, nullptr));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_out = ms * 1000.0f / iters;
    return VT_OK;
} VT_NOTHROW_INT

int vt_op_nv12_to_rgb8_batch_bench(int device_id, int w, int h, int n, int iters, float* us_out) try {
    if (w < 2 || h < 2 || w > 16384 || h > 16384 || n < 1 || n > 1024 || iters < 1 || !us_out) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    const size_t in_bytes = (nv12_bytes_read(w, h) + 16 + 255) & ~(size_t)255, out_bytes = ((size_t)w * h * 3 + 255) & ~(size_t)255;
    DevBuf din, dout;
    HIPCHK(din.alloc(in_bytes * n)); HIPCHK(dout.alloc(out_bytes * n));
    std::vector<uint8_t> host(in_bytes);
    uint32_t seed = 2463534242u;
    for (auto& v : host) { seed ^= seed << 13; seed ^= seed >> 17; seed ^= seed << 5; v = (uint8_t)seed; }
    std::vector<const uint8_t*> ins((size_t)n);
    std::vector<uint8_t*> outs((size_t)n);
    for (int i = 0; i < n; ++i) {
        ins[(size_t)i] = (const uint8_t*)din.p + (size_t)i * in_bytes;
        outs[(size_t)i] = (uint8_t*)dout.p + (size_t)i * out_bytes;
        HIPCHK(hipMemcpy((void*)ins[(size_t)i], host.data(), in_bytes, hipMemcpyHostToDevice));
    }
    for (int i = 0; i < 3; ++i) HIPCHK(launch_nv12_to_rgb8_batch(ins.data(), outs.data(), n, w, h, nullptr));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    HIPCHK(hipEventRecord(e0, nullptr));
    for (int i = 0; i < iters; ++i) HIPCHK(launch_nv12_to_rgb8_batch(ins.data(), outs.data(), n, w, h, nullptr));
    HIPCHK(hipEventRecord(e1, nullptr));
    HIPCHK(hipEventSynchronize(e1));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *us_out = ms * 1000.0f / iters;
    return VT_OK;
} VT_NOTHROW_INT

int vt_op_layernorm(int device_id, const float* x, const float* gamma, const float* beta, float* y, int M, int D) try {
    if (!x || !gamma || !beta || !y || M <= 0 || D % 128) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    DevBuf dx, dg, db, dy;
    HIPCHK(dx.alloc((size_t)M * D * 4)); HIPCHK(dg.alloc((size_t)D * 4)); HIPCHK(db.alloc((size_t)D * 4)); HIPCHK(dy.alloc((size_t)M * D * 2));
    HIPCHK(hipMemcpy(dx.p, x, (size_t)M * D * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dg.p, gamma, (size_t)D * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, beta, (size_t)D * 4, hipMemcpyHostToDevice));
    HIPCHK(launch_layernorm((const float*)dx.p, (const float*)dg.p, (const float*)db.p, (bf16_t*)dy.p, M, D, M, 0, 0, 1e-6f, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<bf16_t> tmp((size_t)M * D);
    HIPCHK(hipMemcpy(tmp.data(), dy.p, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) { uint32_t u = ((uint32_t)tmp[i]) << 16; memcpy(y + i, &u, 4); }
    return VT_OK;
} VT_NOTHROW_INT

// The head's 3x3 convolution (zero padding) + bias + ReLU as the engine runs it: an implicit GEMM over
// t [B*grid*grid][C] (bf16) with w [N][9*C] (bf16, column (ky*3+kx)*C + c); out [B*grid*grid][N] bf16
// widened to f32. cfg 0..3 (4-wave kernel), < 0: the launcher's choice.
int vt_op_conv3x3_relu_bf16(int device_id, const uint16_t* t, const uint16_t* w, const float* bias, float* out,
                            int B, int grid, int C, int N, int cfg) try {
    if (!t || !w || !bias || !out || B < 1 || grid < 1 || C % 64 || N % 64 || cfg > GEMM_CFG_SMALL_MAX)
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(gemm_prepare());
    const size_t M = (size_t)B * grid * grid;
    DevBuf dt, dw, db, dout, dz;
    HIPCHK(dt.alloc(M * C * 2)); HIPCHK(dw.alloc((size_t)N * 9 * C * 2)); HIPCHK(db.alloc((size_t)N * 4));
    HIPCHK(dout.alloc(M * N * 2)); HIPCHK(dz.alloc(256));
    HIPCHK(hipMemcpy(dt.p, t, M * C * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(dw.p, w, (size_t)N * 9 * C * 2, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemset(dz.p, 0, 256));
    GemmArgs g{};
    g.A = (const bf16_t*)dt.p; g.lda = C; g.W = (const bf16_t*)dw.p; g.ldw = 9 * C; g.bias = (const float*)db.p;
    g.M = (int)M; g.N = N; g.K = 9 * C; g.Cb = (bf16_t*)dout.p; g.ldcb = N;
    g.conv_grid = grid; g.conv_C = C; g.zeros = (const bf16_t*)dz.p;
    if (cfg < 0) HIPCHK(launch_gemm(g, EPI_RELU_BF16, nullptr));
    else HIPCHK(launch_gemm_cfg(g, EPI_RELU_BF16, cfg, nullptr));
    HIPCHK(hipDeviceSynchronize());
    std::vector<bf16_t> tmp(M * N);
    HIPCHK(hipMemcpy(tmp.data(), dout.p, tmp.size() * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < tmp.size(); ++i) { uint32_t u = ((uint32_t)tmp[i]) << 16; memcpy(out + i, &u, 4); }
    return VT_OK;
} VT_NOTHROW_INT

// The head's band kernel (k_head.hip) on its own: out = relu(conv(t) + bias), conv3x3 != 0: t [B*grid*grid][Cin],
// w [N][9*Cin], N == Cin; else the 1x1 layer: w [N][Cin]. R / ncb <= 0: the launcher's plan. t == NULL: operands
// filled with a fixed pseudo-random pattern (timing runs). out (nullable): [B*grid*grid][N] bf16 values widened
// to f32. iters > 0 and us_out: mean microseconds per launch over iters launches.
int vt_op_headconv_bf16(int device_id, const uint16_t* t, const uint16_t* w, const float* bias, float* out,
                        int B, int grid, int Cin, int N, int conv3x3, int R, int ncb, int iters, float* us_out) try {
    if (B < 1 || grid < 1 || Cin % 64 || N % 64 || (t && (!w || !bias)))
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    const int K = conv3x3 ? 9 * Cin : Cin;
    if (!headconv_supported(grid, conv3x3 ? Cin : N, N, K, conv3x3 != 0))
        return set_err(VT_ERR_INVALID_ARG, "shape not supported by the band kernel");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(headconv_prepare());
    const size_t M = (size_t)B * grid * grid;
    DevBuf dt, dw, db, dout, dz;
    HIPCHK(dt.alloc(M * Cin * 2)); HIPCHK(dw.alloc((size_t)N * K * 2)); HIPCHK(db.alloc((size_t)N * 4));
    HIPCHK(dout.alloc(M * N * 2)); HIPCHK(dz.alloc(256));
    if (t) {
        HIPCHK(hipMemcpy(dt.p, t, M * Cin * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dw.p, w, (size_t)N * K * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<bf16_t> ht(M * Cin), hw((size_t)N * K);
        uint32_t seed = 777u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (bf16_t)(0x3c00u + ((seed >> 9) & 0x3ffu) + ((seed >> 3) & 0x8000u)); };
        for (auto& v : ht) v = rnd();
        for (auto& v : hw) v = rnd();
        HIPCHK(hipMemcpy(dt.p, ht.data(), ht.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dw.p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(db.p, 0, (size_t)N * 4));
    }
    HIPCHK(hipMemset(dz.p, 0, 256));
    HeadConvArgs h{};
    h.in = (const bf16_t*)dt.p; h.ldin = Cin; h.W = (const bf16_t*)dw.p; h.ldw = K; h.bias = (const float*)db.p;
    h.out = (bf16_t*)dout.p; h.ldout = N; h.zeros = (const bf16_t*)dz.p;
    h.B = B; h.grid = grid; h.C = conv3x3 ? Cin : N; h.N = N; h.K = K; h.conv3x3 = conv3x3 ? 1 : 0;
    h.R = R; h.ncb = ncb;
    HIPCHK(launch_headconv(h, nullptr, nullptr));
    HIPCHK(hipDeviceSynchronize());
    if (out) {
        std::vector<bf16_t> tmp(M * N);
        HIPCHK(hipMemcpy(tmp.data(), dout.p, tmp.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) { uint32_t u = ((uint32_t)tmp[i]) << 16; memcpy(out + i, &u, 4); }
    }
#ifdef VT_STAMPS   // diagnostic builds only: per-wave cycle sums of the main loop, medians printed
    DevBuf ddbg;
    const size_t dbg_words = (size_t)B * grid * 2 * 8 * 4;
    HIPCHK(ddbg.alloc(dbg_words * 8));
    HIPCHK(hipMemset(ddbg.p, 0, dbg_words * 8));
    h.dbg = (unsigned long long*)ddbg.p;
#endif
    if (iters > 0 && us_out) {
        for (int i = 0; i < 3; ++i) HIPCHK(launch_headconv(h, nullptr, nullptr));
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) HIPCHK(launch_headconv(h, nullptr, nullptr));
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *us_out = ms * 1000.0f / iters;
    }
#ifdef VT_STAMPS
    {
        std::vector<unsigned long long> hd(dbg_words);
        HIPCHK(hipMemcpy(hd.data(), ddbg.p, dbg_words * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> col[2][4];
        for (size_t wg = 0; wg < dbg_words / 32; ++wg)
            for (int wv = 0; wv < 8; ++wv) {
                const unsigned long long* d = &hd[(wg * 8 + wv) * 4];
                if (d[3] == 0) continue;
                for (int k = 0; k < 4; ++k) col[wv >= 4][k].push_back(d[k]);
            }
        auto med = [](std::vector<unsigned long long>& v) { if (v.empty()) return 0ull; std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
        fprintf(stderr, "headconv stamps (median cycles per wave over the main loop): computing waves [-, barrier, compute, total] = "
                "%llu %llu %llu %llu; loader waves [vmcnt wait, barrier, issue, total] = %llu %llu %llu %llu\n",
                med(col[0][0]), med(col[0][1]), med(col[0][2]), med(col[0][3]), med(col[1][0]), med(col[1][1]), med(col[1][2]), med(col[1][3]));
    }
#endif
    return VT_OK;
} VT_NOTHROW_INT

// The head's first layer with the final LayerNorm: out[b * ns + cell][n] = ReLU(LayerNorm(xh + xl)[b * ntok + off + cell] . w[n] + bias[n])
// as bf16. fused != 0: one launch (the band kernel normalises its rows itself); fused == 0: the LayerNorm kernel, then the
// band kernel on its output - the form the fused one must reproduce bit for bit. xh == nullptr: synthetic operands (timing).
int vt_op_headconv_ln_bf16(int device_id, const uint16_t* xh, const int8_t* xl, const float* gamma, const float* beta,
                           float eps, int ntok, int off, const uint16_t* w, const float* bias, float* out, int B, int grid,
                           int D, int N, int fused, int R, int ncb, int iters, float* us_out) try {
    const int ns = grid * grid;
    if (B < 1 || grid < 1 || off < 0 || ntok < off + ns || (xh && (!xl || !gamma || !beta || !w || !bias)))
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (!headconv_ln_supported(grid, N, D))
        return set_err(VT_ERR_INVALID_ARG, "shape not supported by the band kernel with the LayerNorm inside");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(headconv_prepare());
    const size_t Mx = (size_t)B * ntok, M = (size_t)B * ns;
    DevBuf dh, dl, dg, dbt, dw, db, dfeat, dout;
    HIPCHK(dh.alloc(Mx * D * 2)); HIPCHK(dl.alloc(Mx * D * 2)); HIPCHK(dg.alloc((size_t)D * 4)); HIPCHK(dbt.alloc((size_t)D * 4));
    HIPCHK(dw.alloc((size_t)N * D * 2)); HIPCHK(db.alloc((size_t)N * 4)); HIPCHK(dfeat.alloc(M * D * 2)); HIPCHK(dout.alloc(M * N * 2));
    if (xh) {
        HIPCHK(hipMemcpy(dh.p, xh, Mx * D * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dl.p, xl, Mx * D, hipMemcpyHostToDevice));       // lo8 bytes
        HIPCHK(hipMemcpy(dg.p, gamma, (size_t)D * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dbt.p, beta, (size_t)D * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dw.p, w, (size_t)N * D * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(db.p, bias, (size_t)N * 4, hipMemcpyHostToDevice));
    } else {
        std::vector<bf16_t> hx(Mx * D), hw((size_t)N * D);
        std::vector<float> ones((size_t)D, 1.0f);
        uint32_t seed = 4242u;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return (bf16_t)(0x3c00u + ((seed >> 9) & 0x3ffu) + ((seed >> 3) & 0x8000u)); };
        for (auto& v : hx) v = rnd();
        for (auto& v : hw) v = rnd();
        HIPCHK(hipMemcpy(dh.p, hx.data(), hx.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(dl.p, 0, Mx * D * 2));
        HIPCHK(hipMemcpy(dg.p, ones.data(), (size_t)D * 4, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(dbt.p, 0, (size_t)D * 4));
        HIPCHK(hipMemcpy(dw.p, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemset(db.p, 0, (size_t)N * 4));
    }
    HeadConvArgs h{};
    h.W = (const bf16_t*)dw.p; h.ldw = D; h.bias = (const float*)db.p; h.out = (bf16_t*)dout.p; h.ldout = N;
    h.B = B; h.grid = grid; h.C = N; h.N = N; h.K = D; h.conv3x3 = 0; h.R = R; h.ncb = ncb;
    if (fused) {
        h.xh = (const bf16_t*)dh.p; h.xl = (const uint8_t*)dl.p; h.ln_g = (const float*)dg.p; h.ln_b = (const float*)dbt.p;
        h.ln_eps = eps; h.in_stride = ntok; h.in_off = off;
    } else {
        h.in = (const bf16_t*)dfeat.p; h.ldin = D;
    }
    auto run = [&]() -> hipError_t {
        if (!fused) {
            hipError_t e = launch_layernorm_split((const bf16_t*)dh.p, (const uint8_t*)dl.p, (const float*)dg.p, (const float*)dbt.p,
                                                  (bf16_t*)dfeat.p, (int)M, D, ns, ntok, off, eps, nullptr);
            if (e != hipSuccess) return e;
        }
        return launch_headconv(h, nullptr, nullptr);
    };
    HIPCHK(run());
    HIPCHK(hipDeviceSynchronize());
    if (out) {
        std::vector<bf16_t> tmp(M * N);
        HIPCHK(hipMemcpy(tmp.data(), dout.p, tmp.size() * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < tmp.size(); ++i) { uint32_t u = ((uint32_t)tmp[i]) << 16; memcpy(out + i, &u, 4); }
    }
    if (iters > 0 && us_out) {
        for (int i = 0; i < 3; ++i) HIPCHK(run());
        hipEvent_t e0, e1;
        HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
        HIPCHK(hipEventRecord(e0, nullptr));
        for (int i = 0; i < iters; ++i) HIPCHK(run());
        HIPCHK(hipEventRecord(e1, nullptr));
        HIPCHK(hipEventSynchronize(e1));
        float ms = 0;
        HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        *us_out = ms * 1000.0f / iters;
    }
    return VT_OK;
} VT_NOTHROW_INT

}  // extern "C"
