// vt_engine.hip — host side of libvittrack_hip.so: weight blob, per-GPU buffers, the per-frame
// launch plan (eager or one hipGraph replay), and the extern "C" ABI of include/vittrack_hip.h.
//
// One Engine = B independent tracked streams on one GPU. Everything a frame needs stays in HBM:
// the decode kernel of frame t writes the box that the preprocessing kernel of frame t+1 reads, so
// a stream of updates is a pure device-side chain; the host only supplies frame pointers and
// collects 24 B of result per stream.
#include "vt_engine.hpp"

static thread_local char g_err[512] = "";
char* vt_err_text() { return g_err; }
int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

thread_local LaunchProbe* vt_launch_probe = nullptr;

// kernel-family label of a GEMM launch (built only when a profiler is attached): the tile configuration
// in it is the one launch_gemm() really runs for these arguments
static const char* gemm_name(int epi, const GemmArgs& a) {
    static const char* tags[] = {"xpos", "xresid", "gelu", "relu", "qkv", "x"};
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "gemm_bf16_%s_%s_n%dk%d", tags[epi], gemm_config_name(gemm_effective_config(a, epi)), a.N, a.K);
    return buf;
}


void Engine::destroy() {
    if (!stream && !d_blob) return;
    (void)hipSetDevice(device);
    if (stream) (void)hipStreamSynchronize(stream);
    drop_graphs();
    void* devp[] = {d_blob, d_patches, d_qk, d_vt, d_attn, d_mlp, d_feat, d_ta, d_tb, d_zeros,
                    d_xh, d_xl, d_cstat, d_rstat, d_panel_cnt, d_band_cnt, d_band_best, d_foldw, d_foldv, d_headout, d_taps, d_states, d_frames,
                    d_results, d_stage};
    for (void* p : devp)
        if (p) (void)hipFree(p);
    if (h_frames) (void)hipHostFree(h_frames);
    if (h_results) (void)hipHostFree(h_results);
    if (h_state) (void)hipHostFree(h_state);
    if (h_pack) (void)hipHostFree(h_pack);
    if (h_states_all) (void)hipHostFree(h_states_all);
    for (HostSlot& sl : hs) {
        if (sl.d_arena) (void)hipFree(sl.d_arena);
        if (sl.h_arena) (void)hipHostFree(sl.h_arena);
        if (sl.h_res) (void)hipHostFree(sl.h_res);
        if (sl.h_st) (void)hipHostFree(sl.h_st);
        if (sl.up_ev) (void)hipEventDestroy(sl.up_ev);
        if (sl.done_ev) (void)hipEventDestroy(sl.done_ev);
        sl = HostSlot();
    }
    if (copy_stream) { (void)hipStreamSynchronize(copy_stream); (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
    for (int i = 0; i < RING; ++i)
        if (ring_ev[i]) (void)hipEventDestroy(ring_ev[i]);
    if (stream) (void)hipStreamDestroy(stream);
    stream = nullptr;
    d_blob = nullptr;
}

void Engine::drop_graphs() {
    for (int t = 0; t < TIERS; ++t) {
        if (graph_exec[t]) (void)hipGraphExecDestroy(graph_exec[t]);
        if (graph[t]) (void)hipGraphDestroy(graph[t]);
        graph_exec[t] = nullptr; graph[t] = nullptr;
    }
}

// The crop kernel's buffer tier for the pass about to be enqueued: the largest any stream's last known box needs (the
// boxes of a pipelined pass are one pass old: targets change by a few per cent per frame, the tier has headroom, and a
// tile that does not fit its buffer after all takes the per-pixel path - slower, never wrong).
int Engine::pick_crop_tier() const {
    if (crop_tier_forced >= 0) return std::min(crop_tier_forced, TIERS - 1);
    int t = 0;
    for (int b = 0; b < B && t < TIERS - 1; ++b)
        t = std::max(t, std::min(preproc_tier_for_box(d, known[b].box[2], known[b].box[3], false), TIERS - 1));
    return t;
}

int Engine::index_blob(const uint8_t* hc, size_t bytes) {
    if (bytes < kHeaderBytes || memcmp(hc, kMagic, 8) != 0)
        return set_err(VT_ERR_FORMAT, "weight blob: bad magic or truncated header");
    int32_t ints[20];
    float fl[8];
    memcpy(ints, hc + 8, sizeof(ints));
    memcpy(fl, hc + 8 + 80, sizeof(fl));
    if (ints[0] != 1) return set_err(VT_ERR_FORMAT, "weight blob: unsupported version %d", ints[0]);
    d.patch = ints[1]; d.T = ints[2]; d.S = ints[3]; d.D = ints[4]; d.H = ints[5]; d.L = ints[6];
    d.mlp = ints[7]; d.C = ints[8]; d.kpad = ints[9];
    const int n_tensors = ints[10];
    for (int i = 0; i < 3; ++i) { d.norm_a[i] = fl[i]; d.norm_b[i] = fl[3 + i]; }
    d.success_threshold = fl[6];
    d.ln_eps = fl[7];
    // every dimension is bounded BEFORE anything is derived from it (a corrupt or crafted blob must
    // not overflow the int arithmetic below or make layers.resize() throw)
    if (d.patch < 2 || d.patch > 64 || d.T < d.patch || d.S < d.patch || d.T > 4096 || d.S > 4096 ||
        d.T % d.patch || d.S % d.patch)
        return set_err(VT_ERR_FORMAT, "weight blob: patch %d / template %d / search %d out of range or "
                       "not multiples of the patch", d.patch, d.T, d.S);
    if (d.L < 1 || d.L > 64 || d.D < 128 || d.D > 1536 || d.mlp < 64 || d.mlp > 16384 || d.C < 64 ||
        d.C > 1024 || d.kpad < 64 || d.kpad > 16384)
        return set_err(VT_ERR_FORMAT, "weight blob: model dimensions out of range (layers=%d D=%d "
                       "mlp=%d C=%d kpad=%d)", d.L, d.D, d.mlp, d.C, d.kpad);
    d.gt = d.T / d.patch; d.gs = d.S / d.patch;
    d.nt = d.gt * d.gt; d.ns = d.gs * d.gs; d.ntok = d.nt + d.ns;
    d.npad = (d.ntok + 63) / 64 * 64;
    if (d.ntok > 16384)
        return set_err(VT_ERR_FORMAT, "weight blob: %d tokens per frame (limit 16384)", d.ntok);
    {   // widths the LayerNorm kernels are instantiated for (k_misc.hip launch_layernorm)
        const int q = d.D / 128;
        const bool ln_ok = d.D % 128 == 0 && (q <= 4 || q == 6 || q == 8 || q == 10 || q == 12);
        if (!ln_ok)
            return set_err(VT_ERR_FORMAT, "weight blob: embedding width D=%d is not supported (LayerNorm "
                           "kernels exist for D in {128, 256, 384, 512, 768, 1024, 1280, 1536})", d.D);
    }
    if (d.H != d.D / 64 || d.mlp % 64 || d.C % 64 || d.kpad % 64 ||
        d.kpad < 3 * d.patch * d.patch || (d.ntok & 3) || (d.ns & 3))
        return set_err(VT_ERR_FORMAT, "weight blob: unsupported model shape (D=%d H=%d mlp=%d C=%d "
                       "kpad=%d tokens=%d)", d.D, d.H, d.mlp, d.C, d.kpad, d.ntok);
    if (n_tensors <= 0 || n_tensors > 4096 || kHeaderBytes + (size_t)n_tensors * kEntryBytes > bytes)
        return set_err(VT_ERR_FORMAT, "weight blob: tensor table out of range");
    const uint64_t table_end = kHeaderBytes + (uint64_t)n_tensors * kEntryBytes;
    tens.clear();
    for (int i = 0; i < n_tensors; ++i) {
        BlobEntry e;
        memcpy(&e, hc + kHeaderBytes + (size_t)i * kEntryBytes, sizeof(e));
        e.name[31] = 0;
        const uint64_t esz = e.dtype == 1 ? 2 : 4;
        // offset/nbytes checked without forming offset + nbytes (which wraps for offset near 2^64);
        // data may not overlap the header or the table
        if (e.dtype > 1 || e.offset % 16 || e.offset < table_end || e.offset > bytes ||
            e.nbytes > bytes - e.offset || e.rows == 0 || e.cols == 0 || e.rows > (1u << 20) ||
            e.cols > (1u << 20) || e.nbytes != (uint64_t)e.rows * e.cols * esz)
            return set_err(VT_ERR_FORMAT, "weight blob: tensor '%s' malformed", e.name);
        TensorRef r;
        r.ptr = d_blob + e.offset;
        r.dtype = e.dtype; r.rows = e.rows; r.cols = e.cols;
        tens[e.name] = r;
    }
    auto need = [&](const std::string& n, uint32_t dt, uint32_t rows, uint32_t cols) -> const void* {
        const TensorRef* t = find(n);
        if (!t || t->dtype != dt || t->rows != rows || t->cols != cols) {
            set_err(VT_ERR_FORMAT, "weight blob: tensor '%s' missing or wrong shape", n.c_str());
            return nullptr;
        }
        return t->ptr;
    };
    const uint32_t D = d.D, C = d.C;
    if (!need("patch_w", 1, D, d.kpad) || !need("patch_b", 0, 1, D) || !need("pos", 0, d.ntok, D) ||
        !need("norm_g", 0, 1, D) || !need("norm_b", 0, 1, D) || !need("head.w0", 1, C, D) ||
        !need("head.b0", 0, 1, C) || !need("head.w1", 1, C, 9 * C) || !need("head.b1", 0, 1, C) ||
        !need("head.w2", 1, C, 9 * C) || !need("head.b2", 0, 1, C) ||
        !need("head.w3", 1, C, 9 * C) || !need("head.b3", 0, 1, C) || !need("head.w4", 0, 8, C) ||
        !need("head.b4", 0, 1, 8) || !need("hann", 0, 1, d.ns))
        return VT_ERR_FORMAT;
    layers.resize(d.L);
    for (int l = 0; l < d.L; ++l) {
        const std::string p = "l" + std::to_string(l) + ".";
        LayerW& w = layers[l];
        w.ln1_g = (const float*)need(p + "ln1_g", 0, 1, D);
        w.ln1_b = (const float*)need(p + "ln1_b", 0, 1, D);
        w.qkv_w = (const bf16_t*)need(p + "qkv_w", 1, 3 * D, D);
        w.qkv_b = (const float*)need(p + "qkv_b", 0, 1, 3 * D);
        w.proj_w = (const bf16_t*)need(p + "proj_w", 1, D, D);
        w.proj_b = (const float*)need(p + "proj_b", 0, 1, D);
        w.ln2_g = (const float*)need(p + "ln2_g", 0, 1, D);
        w.ln2_b = (const float*)need(p + "ln2_b", 0, 1, D);
        w.fc1_w = (const bf16_t*)need(p + "fc1_w", 1, d.mlp, D);
        w.fc1_b = (const float*)need(p + "fc1_b", 0, 1, d.mlp);
        w.fc2_w = (const bf16_t*)need(p + "fc2_w", 1, D, d.mlp);
        w.fc2_b = (const float*)need(p + "fc2_b", 0, 1, D);
        if (!w.ln1_g || !w.ln1_b || !w.qkv_w || !w.qkv_b || !w.proj_w || !w.proj_b || !w.ln2_g ||
            !w.ln2_b || !w.fc1_w || !w.fc1_b || !w.fc2_w || !w.fc2_b)
            return VT_ERR_FORMAT;
    }
    return VT_OK;
}

int Engine::load_blob_host(const std::vector<uint8_t>& blob) {
    blob_bytes = blob.size();
    HIPCHK(hipMalloc((void**)&d_blob, blob_bytes));
    HIPCHK(hipMemcpy(d_blob, blob.data(), blob_bytes, hipMemcpyHostToDevice));
    return index_blob(blob.data(), blob_bytes);
}

int Engine::load_blob_device(const void* d_src, size_t bytes) {
    if (bytes < kHeaderBytes) return set_err(VT_ERR_FORMAT, "weight blob: truncated");
    std::vector<uint8_t> head(kHeaderBytes);
    HIPCHK(hipMemcpy(head.data(), d_src, kHeaderBytes, hipMemcpyDeviceToHost));
    if (memcmp(head.data(), kMagic, 8) != 0) return set_err(VT_ERR_FORMAT, "weight blob: bad magic");
    int32_t n_tensors;
    memcpy(&n_tensors, head.data() + 8 + 10 * 4, 4);
    const size_t tbl = kHeaderBytes + (size_t)std::max(n_tensors, 0) * kEntryBytes;
    if (n_tensors <= 0 || tbl > bytes) return set_err(VT_ERR_FORMAT, "weight blob: bad table");
    // host copy of header + table only; index_blob checks offsets against the full size
    std::vector<uint8_t> hc(tbl);
    HIPCHK(hipMemcpy(hc.data(), d_src, tbl, hipMemcpyDeviceToHost));
    blob_bytes = bytes;
    HIPCHK(hipMalloc((void**)&d_blob, blob_bytes));
    HIPCHK(hipMemcpy(d_blob, d_src, blob_bytes, hipMemcpyDeviceToDevice));
    HIPCHK(hipStreamSynchronize(nullptr));      // a device-to-device hipMemcpy may return early; the engine's
                                                // stream (non-blocking) is not ordered behind the null stream
    hc.resize(tbl);
    // index_blob only touches [0, tbl) of the host copy
    return index_blob(hc.data(), blob_bytes);
}


// HBM the activations of B streams need (bytes), as alloc_buffers() lays them out
size_t Engine::activation_bytes() const {
    const size_t M = (size_t)B * d.ntok, Ms = (size_t)B * d.ns;
    const size_t fold_rows = (size_t)d.L * (3 * d.D + d.mlp);      // folded QKV + fc1 weights of every layer
    return 2 * (M * d.kpad + M * d.D + M * 2 * d.D + (size_t)B * d.H * 64 * d.npad + M * d.D + M * d.mlp +
                Ms * d.D + 2 * Ms * d.C + fold_rows * d.D) + M * d.D /* lo8 plane */ +
           8 * (M * (d.D / VT_STAT_CHUNK) + M) + 4 * (Ms * 8 + 2 * fold_rows) +
           (size_t)B * (sizeof(StreamState) + sizeof(FrameDesc) + sizeof(vt_result));
}

int Engine::alloc_buffers() {
    const size_t M = (size_t)B * d.ntok, Ms = (size_t)B * d.ns;
    {   // fail early and cleanly (VT_ERR_OOM) instead of half-way through a dozen hipMallocs
        const size_t need = activation_bytes();
        if (max_device_bytes && need + blob_bytes > max_device_bytes)
            return set_err(VT_ERR_OOM, "%d streams need %.1f MiB of HBM (+ %.1f MiB of weights); "
                           "vt_config.max_device_mib allows %.1f", B, need / 1048576.0,
                           blob_bytes / 1048576.0, max_device_bytes / 1048576.0);
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need > free_b)
            return set_err(VT_ERR_OOM, "%d streams need %.1f MiB of HBM for activations; %.1f MiB "
                           "are free on device %d", B, need / 1048576.0, free_b / 1048576.0, device);
    }
    HIPCHK(dalloc0(&d_patches, M * d.kpad, stream));
    HIPCHK(dalloc0(&d_xh, M * d.D, stream));
    HIPCHK(dalloc0(&d_xl, M * d.D, stream));       // bytes
    HIPCHK(dalloc0(&d_cstat, M * (d.D / VT_STAT_CHUNK), stream));
    HIPCHK(dalloc0(&d_rstat, M + 1, stream));      // + 1: the 4-wave kernel fetches row terms as aligned pairs
    HIPCHK(dalloc0(&d_panel_cnt, (M + 255) / 256 + 1, stream));
    {   // fold LayerNorm 1 / 2 of every layer into the QKV / fc1 weights
        const size_t rows = (size_t)3 * d.D + d.mlp;
        HIPCHK(dalloc0(&d_foldw, (size_t)d.L * rows * d.D, stream));
        HIPCHK(dalloc0(&d_foldv, (size_t)d.L * 2 * rows, stream));
        for (int l = 0; l < d.L; ++l) {
            LayerW& w = layers[l];
            bf16_t* fw = d_foldw + (size_t)l * rows * d.D;
            float* fv = d_foldv + (size_t)l * 2 * rows;
            HIPCHK(launch_fold_layernorm(w.qkv_w, w.ln1_g, w.ln1_b, w.qkv_b, fw, fv, fv + rows, 3 * d.D, d.D, stream));
            HIPCHK(launch_fold_layernorm(w.fc1_w, w.ln2_g, w.ln2_b, w.fc1_b, fw + (size_t)3 * d.D * d.D, fv + 3 * d.D,
                                         fv + rows + 3 * d.D, d.mlp, d.D, stream));
            w.qkv_wf = fw; w.qkv_cs = fv; w.qkv_c = fv + rows;
            w.fc1_wf = fw + (size_t)3 * d.D * d.D; w.fc1_cs = fv + 3 * d.D; w.fc1_c = fv + rows + 3 * d.D;
        }
        HIPCHK(hipStreamSynchronize(stream));
    }
    HIPCHK(dalloc0(&d_qk, M * 2 * d.D, stream));
    HIPCHK(dalloc0(&d_vt, (size_t)B * d.H * 64 * d.npad, stream));
    HIPCHK(dalloc0(&d_attn, M * d.D, stream));
    HIPCHK(dalloc0(&d_mlp, M * d.mlp, stream));
    HIPCHK(dalloc0(&d_feat, Ms * d.D, stream));
    HIPCHK(dalloc0(&d_ta, Ms * d.C, stream));
    HIPCHK(dalloc0(&d_tb, Ms * d.C, stream));
    HIPCHK(dalloc0(&d_zeros, (size_t)128, stream));
    HIPCHK(dalloc0(&d_headout, Ms * 8, stream));
    HIPCHK(dalloc0(&d_states, (size_t)B, stream));
    HIPCHK(dalloc0(&d_band_cnt, (size_t)B + 1, stream));
    HIPCHK(dalloc0(&d_band_best, (size_t)B * d.gs * 2, stream));
    {   // B frame descriptors + the pass's PassOut behind them (one upload per pass)
        void* p = nullptr;
        HIPCHK(hipMalloc(&p, frames_block_bytes()));
        HIPCHK(hipMemsetAsync(p, 0, frames_block_bytes(), stream));
        d_frames = (FrameDesc*)p;
    }
    HIPCHK(dalloc0(&d_results, (size_t)B, stream));
    HIPCHK(hipHostMalloc((void**)&h_frames, frames_block_bytes() * RING));
    HIPCHK(hipHostMalloc((void**)&h_results, sizeof(vt_result) * B));
    HIPCHK(hipHostMalloc((void**)&h_state, sizeof(StreamState)));
    HIPCHK(hipHostMalloc((void**)&h_states_all, sizeof(StreamState) * B));
    memset(h_states_all, 0, sizeof(StreamState) * B);
    memset(h_results, 0, sizeof(vt_result) * B);
    for (int i = 0; i < RING; ++i) HIPCHK(hipEventCreateWithFlags(&ring_ev[i], hipEventDisableTiming));
    h_initialized.assign(B, 0);
    known.assign((size_t)B, StreamState{});
    HIPCHK(hipStreamSynchronize(stream));       // every fill has landed before the handle is handed out
    return VT_OK;
}

double Engine::flops_encoder() const {
    const double n = d.ntok, D = d.D;
    const double per_layer = 2 * n * D * 3 * D + 2 * n * D * D + 4 * n * D * d.mlp + 4 * n * n * D;
    return d.L * per_layer + 2 * n * (3.0 * d.patch * d.patch) * D;
}
double Engine::flops_head() const {
    const double C = d.C;
    return 2.0 * d.ns * (d.D * C + 27 * C * C + 8 * C);
}

// One hot-path pass over all B streams. With prof != nullptr every launch is bracketed by HIP
// events on this engine's stream.
int Engine::run_pass(Profiler* prof) {
    const int M = B * d.ntok, Ms = B * d.ns, D = d.D;
    hipError_t lerr = hipSuccess;
    auto L = [&](const char* name, double flops, double bytes, auto&& fn) {
        if (lerr != hipSuccess) return;
        if (prof) {
            Profiler::Rec r;
            r.fam = prof->family(name);
            (void)hipEventCreate(&r.a);
            (void)hipEventCreate(&r.b);
            (void)hipEventCreate(&r.k.start);
            (void)hipEventCreate(&r.k.stop);
            r.k.launches = 0;
            (void)hipEventRecord(r.a, stream);
            vt_launch_probe = &r.k;
            lerr = fn();
            vt_launch_probe = nullptr;
            (void)hipEventRecord(r.b, stream);
            prof->recs.push_back(r);
            prof->fams[r.fam].launches += 1;
            prof->fams[r.fam].flops += flops;
            prof->fams[r.fam].bytes += bytes;
        } else {
            lerr = fn();
        }
    };
    auto gemm = [&](int epi, GemmArgs a) {
        const double fl = 2.0 * a.M * a.N * a.K;
        // algorithmic bytes: operands once, output once; the 3-byte residual pair is read and written (3 + 3 B)
        const double by = 2.0 * ((double)a.M * a.K + (double)a.N * a.K) +
                          (epi == EPI_RESID ? 6.0 : epi == EPI_F32_POS ? 3.0 : 2.0) * a.M * a.N;
        L(prof ? gemm_name(epi, a) : "", fl, by, [&] { return launch_gemm(a, epi, stream); });
    };
    auto tap = [&](int slot) {
        if (taps && lerr == hipSuccess) {     // both halves of the residual stream: [slot][hi | lo][M][D]
            uint8_t* dst = d_taps + (size_t)slot * tap_slot_bytes();
            lerr = hipMemcpyAsync(dst, d_xh, sizeof(bf16_t) * M * D, hipMemcpyDeviceToDevice, stream);
            if (lerr == hipSuccess)
                lerr = hipMemcpyAsync(dst + sizeof(bf16_t) * M * D, d_xl, (size_t)M * D, hipMemcpyDeviceToDevice, stream);
        }
    };
    const int nchunk = D / VT_STAT_CHUNK;
    // An X-epilogue GEMM (writes the residual pair) followed by the row terms (rstd, -mean * rstd) of the
    // LayerNorm that consumes it: finalized inside the GEMM by the last workgroup of every row panel (the
    // 256x256 kernel), else by a small launch of their own from the chunk partials
    // ... or, where the consumer runs on the 4-wave kernel (few streams), by the consumer's own epilogue:
    // `consumer` (the GEMM with the folded LayerNorm, arguments complete but for the row terms) gets
    // rowstat or cstat_in set accordingly.
    auto xgemm = [&](int epi, GemmArgs a, GemmArgs* consumer, int consumer_epi) {
        const bool stats = consumer != nullptr;
        a.Xh = d_xh; a.Xl = d_xl; a.ldx = D;
        a.cstat = stats ? d_cstat : nullptr;
        a.rowstat_out = stats ? d_rstat : nullptr;
        a.panel_cnt = d_panel_cnt;
        a.ln_eps = d.ln_eps;
        const bool fused = stats && gemm_finalizes_rowstat(a, epi);
        if (!fused) a.rowstat_out = nullptr;
        gemm(epi, a);
        if (!stats) return;
        consumer->ln_eps = d.ln_eps;
        if (!fused && gemm_effective_config(*consumer, consumer_epi) <= GEMM_CFG_SMALL_MAX && consumer->K <= 1024 && consumer->K % 128 == 0) {
            consumer->cstat_in = d_cstat;       // combined in the consumer's epilogue
            return;
        }
        consumer->rowstat = d_rstat;
        if (!fused)
            L("rowstat", 0, (double)M * (nchunk + 1) * 8,
              [&] { return launch_rowstat_finalize(d_cstat, d_rstat, M, nchunk, d.ln_eps, stream); });
    };
    auto qkv_args = [&](int l) {
        const LayerW& w = layers[l];
        GemmArgs a{};
        a.A = d_xh; a.lda = D; a.W = w.qkv_wf; a.ldw = D; a.bias = w.qkv_c; a.colsum = w.qkv_cs;
        a.M = M; a.N = 3 * D; a.K = D;
        a.qk = d_qk; a.vt = d_vt; a.tokens = d.ntok; a.npad = d.npad; a.D = D;
        a.vt_perm = attention_vt_perm(attention_pick_mode(d.ntok, d.npad));   // layout the attention kernel reads
        return a;
    };
    GemmArgs qkv = qkv_args(0);             // LayerNorm 1 is folded into the QKV GEMM

    // K1: crop + resize + normalise the search window of every stream -> patch rows
    L("preproc_search", 0, (double)B * (d.S * d.S * 3 * 2 + 1.5 * d.S * d.S),
      [&] { return launch_preproc(d_frames, d_states, d_patches, d, 0, B, false, stream, crop_tier); });
    // K2: patch embedding (+bias +pos) -> residual stream (3-byte pair + chunk statistics)
    {
        GemmArgs a{};
        a.A = d_patches; a.lda = d.kpad;
        a.W = (const bf16_t*)find("patch_w")->ptr; a.ldw = d.kpad;
        a.bias = (const float*)find("patch_b")->ptr;
        a.M = M; a.N = D; a.K = d.kpad;
        a.pos = (const float*)find("pos")->ptr; a.pos_rows = d.ntok;
        xgemm(EPI_F32_POS, a, &qkv, EPI_QKV);       // + the row terms of block 0's LayerNorm 1
    }
    tap(0);
    for (int l = 0; l < d.L; ++l) {
        const LayerW& w = layers[l];
        gemm(EPI_QKV, qkv);
        L("attention", 4.0 * B * (double)d.ntok * d.ntok * D, (double)M * D * 8, [&] {
            return launch_attention(d_qk, d_vt, d_attn, B, d.ntok, d.H, d.npad, stream);
        });
        {
            GemmArgs a{};
            a.A = d_attn; a.lda = D; a.W = w.proj_w; a.ldw = D; a.bias = w.proj_b;
            a.M = M; a.N = D; a.K = D;
            GemmArgs f{};                   // LayerNorm 2 is folded into fc1
            f.A = d_xh; f.lda = D; f.W = w.fc1_wf; f.ldw = D; f.bias = w.fc1_c; f.colsum = w.fc1_cs;
            f.M = M; f.N = d.mlp; f.K = D; f.Cb = d_mlp; f.ldcb = d.mlp;
            xgemm(EPI_RESID, a, &f, EPI_GELU_BF16);      // + the row terms of LayerNorm 2
            gemm(EPI_GELU_BF16, f);
        }
        {
            GemmArgs a{};
            a.A = d_mlp; a.lda = d.mlp; a.W = w.fc2_w; a.ldw = d.mlp; a.bias = w.fc2_b;
            a.M = M; a.N = D; a.K = d.mlp;
            if (l + 1 < d.L) {              // + the next block's LayerNorm 1 (the final LayerNorm reads the rows itself)
                qkv = qkv_args(l + 1);
                xgemm(EPI_RESID, a, &qkv, EPI_QKV);
            } else {
                xgemm(EPI_RESID, a, nullptr, 0);
            }
        }
        tap(1 + l);
    }
    // final LayerNorm on the search tokens only, compacted to [B*ns][D] - as a launch of its own unless the head's
    // first layer normalises its rows itself (k_head.hip, LNC)
    const bool band = head_band_kernel && head_band_ok;
    const bool ln_fused = band && head_band_kernel >= 2 && headconv_ln_supported(d.gs, d.C, D);
    feat_in_head = ln_fused;
    if (!ln_fused)
        L("layernorm", 0, (double)Ms * D * 6, [&] { return final_layernorm(); });
    // centre head: 1x1 conv, three 3x3 convs, then the f32 5-logit layer + decode. On the band kernel of
    // k_head.hip (the A image of a band resident in LDS, logits + decode fused behind the last layer: 4 launches)
    // where the shape allows it, else as implicit GEMMs on the 4-wave kernel + head_out + decode (6 launches).
    DecodeArgs dec{};
    dec.w4 = (const float*)find("head.w4")->ptr;
    dec.b4 = (const float*)find("head.b4")->ptr;
    dec.hann = (const float*)find("hann")->ptr;
    dec.head_out = d_headout; dec.states = d_states; dec.results = d_results;
    dec.out = (const PassOut*)(d_frames + B);
    dec.B = B; dec.ns = d.ns; dec.grid = d.gs; dec.C = d.C;
    dec.success_threshold = success_threshold;
    bf16_t* cur = d_ta;
    bf16_t* nxt = d_tb;
    if (band) {
        HeadConvArgs h{};
        h.in = d_feat; h.ldin = D; h.W = (const bf16_t*)find("head.w0")->ptr; h.ldw = D;
        h.bias = (const float*)find("head.b0")->ptr; h.out = d_ta; h.ldout = d.C; h.zeros = d_zeros;
        h.B = B; h.grid = d.gs; h.C = d.C; h.N = d.C; h.K = D; h.conv3x3 = 0;
        if (ln_fused) {
            h.in = nullptr;
            h.xh = d_xh; h.xl = d_xl; h.ln_g = (const float*)find("norm_g")->ptr; h.ln_b = (const float*)find("norm_b")->ptr;
            h.ln_eps = d.ln_eps; h.in_stride = d.ntok; h.in_off = d.nt;
        }
        L(prof ? (ln_fused ? "head_ln_conv1x1" : "head_conv1x1") : "", 2.0 * Ms * d.C * D,
          2.0 * ((double)Ms * D * (ln_fused ? 2 : 1) + (double)d.C * D + (double)Ms * d.C),
          [&] { return launch_headconv(h, nullptr, stream); });
        for (int k = 1; k <= 3; ++k) {
            const std::string wn = "head.w" + std::to_string(k), bn = "head.b" + std::to_string(k);
            HeadConvArgs c{};
            c.in = cur; c.ldin = d.C; c.W = (const bf16_t*)find(wn)->ptr; c.ldw = 9 * d.C;
            c.bias = (const float*)find(bn)->ptr; c.out = nxt; c.ldout = d.C; c.zeros = d_zeros;
            c.B = B; c.grid = d.gs; c.C = d.C; c.N = d.C; c.K = 9 * d.C; c.conv3x3 = 1;
            c.band_cnt = d_band_cnt; c.band_best = d_band_best;
            const bool tail = k == 3;
            const double fl = 2.0 * Ms * d.C * 9.0 * d.C + (tail ? 2.0 * Ms * d.C * 5 : 0.0);
            const double by = 2.0 * (2.0 * Ms * d.C + 9.0 * d.C * d.C);
            L(prof ? (tail ? "head_conv3x3_logits_decode" : "head_conv3x3") : "", fl, by,
              [&] { return launch_headconv(c, tail ? &dec : nullptr, stream); });
            std::swap(cur, nxt);
        }
    } else {
        {
            GemmArgs a{};
            a.A = d_feat; a.lda = D; a.W = (const bf16_t*)find("head.w0")->ptr; a.ldw = D;
            a.bias = (const float*)find("head.b0")->ptr;
            a.M = Ms; a.N = d.C; a.K = D; a.Cb = d_ta; a.ldcb = d.C;
            gemm(EPI_RELU_BF16, a);
        }
        for (int k = 1; k <= 3; ++k) {     // 3x3 convs as implicit GEMMs: the im2col row is gathered by the A loads
            GemmArgs a{};
            const std::string wn = "head.w" + std::to_string(k), bn = "head.b" + std::to_string(k);
            a.A = cur; a.lda = d.C; a.W = (const bf16_t*)find(wn)->ptr; a.ldw = 9 * d.C;
            a.bias = (const float*)find(bn)->ptr;
            a.M = Ms; a.N = d.C; a.K = 9 * d.C; a.Cb = nxt; a.ldcb = d.C;
            a.conv_grid = d.gs; a.conv_C = d.C; a.zeros = d_zeros;
            gemm(EPI_RELU_BF16, a);
            std::swap(cur, nxt);
        }
        dec.t3 = cur;
        L("decode", 2.0 * Ms * d.C * 5, (double)Ms * d.C * 2, [&] { return launch_decode(dec, stream); });
    }
    if (lerr != hipSuccess)
        return set_err(VT_ERR_HIP, "kernel launch failed: %s", hipGetErrorString(lerr));
    return VT_OK;     // results and states reach the host through the decode kernel's own stores (PassOut)
}

hipError_t Engine::final_layernorm() {
    return launch_layernorm_split(d_xh, d_xl, (const float*)find("norm_g")->ptr, (const float*)find("norm_b")->ptr, d_feat,
                                  (int)((size_t)B * d.ns), d.D, d.ns, d.ntok, d.nt, d.ln_eps, stream);
}

int Engine::capture_graph(int tier) {
    crop_tier = tier;
    HIPCHK(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
    int rc = run_pass(nullptr);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(stream, &g);
    if (rc != VT_OK) {
        if (g) (void)hipGraphDestroy(g);
        return rc;
    }
    if (e != hipSuccess) return set_err(VT_ERR_HIP, "hipStreamEndCapture: %s", hipGetErrorString(e));
    hipGraphExec_t x = nullptr;
    e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
    if (e != hipSuccess) {          // nothing half-built stays behind: the next attempt starts from scratch
        (void)hipGraphDestroy(g);
        return set_err(e == hipErrorOutOfMemory ? VT_ERR_OOM : VT_ERR_HIP, "hipGraphInstantiate: %s", hipGetErrorString(e));
    }
    graph[tier] = g;
    graph_exec[tier] = x;
    graph_captures += 1;
    return VT_OK;
}

// Every crop-buffer tier's pass, captured and instantiated NOW (engine creation, vt_group_set_tuning): the hot path only
// replays. A live stream whose target grows across a tier boundary (~130 / ~200 px at search 384) must not pay a
// capture + instantiate inside an update (60 fps: /root/reference/src/pipeline.rs:26-37).
int Engine::capture_all_graphs() {
    if (!use_graph) return VT_OK;
    const int keep = crop_tier;
    for (int t = 0; t < TIERS; ++t)
        if (!graph_exec[t])
            if (int rc = capture_graph(t)) { crop_tier = keep; return rc; }
    crop_tier = keep;
    return VT_OK;
}

int check_frame(const vt_frame& f) {
    if (!f.plane0 || f.width < 16 || f.height < 16 || f.width > 16384 || f.height > 16384)
        return set_err(VT_ERR_INVALID_ARG, "frame: null plane or size out of range");
    if (f.format != VT_PIX_RGB8 && f.format != VT_PIX_NV12 && f.format != VT_PIX_YUY2)
        return set_err(VT_ERR_INVALID_ARG, "unknown pixel format %d", f.format);
    const bool window = f.origin_x != 0 || f.origin_y != 0 || f.windowed == 1;
    if (f.origin_x < 0 || f.origin_y < 0 || f.origin_x >= f.width || f.origin_y >= f.height ||
        (f.format == VT_PIX_NV12 && ((f.origin_x | f.origin_y) & 1)) ||
        (f.format == VT_PIX_YUY2 && (f.origin_x & 1)))
        return set_err(VT_ERR_INVALID_ARG, "frame window origin %d,%d invalid", f.origin_x, f.origin_y);
    // extent of what the planes hold: the kernels never read outside it (fetch_rgb, k_preproc.hip)
    int ww = f.width, wh = f.height;
    (void)wh;
    if (window) {
        if (f.window_w < 1 || f.window_h < 1 || f.window_w > f.width - f.origin_x ||
            f.window_h > f.height - f.origin_y)
            return set_err(VT_ERR_INVALID_ARG, "windowed frame needs window_w/window_h inside the frame "
                           "(got %dx%d at %d,%d of %dx%d)", f.window_w, f.window_h, f.origin_x, f.origin_y,
                           f.width, f.height);
        ww = f.window_w; wh = f.window_h;
        if (f.format == VT_PIX_NV12 && (((ww & 1) && f.origin_x + ww != f.width) ||
                                        ((wh & 1) && f.origin_y + wh != f.height)))
            return set_err(VT_ERR_INVALID_ARG, "nv12 window extent must be even unless it ends at the frame edge");
    } else if (f.window_w != 0 || f.window_h != 0) {
        if (f.window_w != f.width || f.window_h != f.height)
            return set_err(VT_ERR_INVALID_ARG, "window_w/window_h set on a frame that is not windowed");
    }
    if (f.format == VT_PIX_RGB8) {
        if (f.stride0 < ww * 3) return set_err(VT_ERR_INVALID_ARG, "rgb8 stride < 3*width");
    } else if (f.format == VT_PIX_NV12) {
        if (!f.plane1 || f.stride0 < ww || f.stride1 < ((ww + 1) & ~1))
            return set_err(VT_ERR_INVALID_ARG, "nv12: null UV plane or stride too small");
    } else {
        if ((f.width & 1) || f.stride0 < ((ww + 1) & ~1) * 2)
            return set_err(VT_ERR_INVALID_ARG, "yuy2: odd width or stride < 2*width");
    }
    return VT_OK;
}

void to_desc(const vt_frame& f, FrameDesc* o) {
    const bool window = f.origin_x != 0 || f.origin_y != 0 || f.windowed == 1;
    o->p0 = (const uint8_t*)f.plane0;
    o->p1 = (const uint8_t*)f.plane1;
    o->w = f.width; o->h = f.height; o->s0 = f.stride0; o->s1 = f.stride1; o->fmt = f.format;
    o->x0 = f.origin_x; o->y0 = f.origin_y;
    o->ww = window ? f.window_w : f.width;
    o->wh = window ? f.window_h : f.height;
    o->pad = 0;
}

int Engine::init_stream(int b, const vt_frame* f, vt_bbox box) {
    if (b < 0 || b >= B || !f) return set_err(VT_ERR_INVALID_ARG, "init: bad stream index");
    if (int rc = check_frame(*f)) return rc;
    if (box.width < 1 || box.height < 1 || box.width > 32768 || box.height > 32768 ||
        box.x < -32768 || box.y < -32768 || box.x > 32768 || box.y > 32768)
        return set_err(VT_ERR_INVALID_ARG, "init: bbox %d,%d %dx%d out of range", box.x, box.y,
                       box.width, box.height);
    DEVICE_SCOPE(device);
    HIPCHK(hipStreamSynchronize(stream));
    memset(h_state, 0, sizeof(StreamState));
    h_state->box[0] = (float)box.x; h_state->box[1] = (float)box.y;
    h_state->box[2] = (float)box.width; h_state->box[3] = (float)box.height;
    h_state->frame_w = f->width; h_state->frame_h = f->height;
    h_state->initialized = 1;
    HIPCHK(hipMemcpyAsync(d_states + b, h_state, sizeof(StreamState), hipMemcpyHostToDevice, stream));
    FrameDesc* slot = h_frames;  // stream is idle: ring slot 0 is free
    to_desc(*f, slot);
    HIPCHK(hipMemcpyAsync(d_frames + b, slot, sizeof(FrameDesc), hipMemcpyHostToDevice, stream));
    HIPCHK(launch_preproc(d_frames, d_states, d_patches, d, b, 1, true, stream,
                          preproc_tier_for_box(d, (float)box.width, (float)box.height, true)));
    HIPCHK(hipStreamSynchronize(stream));
    h_states_all[b] = *h_state;
    known[b] = *h_state;
    h_initialized[b] = 1;
    return VT_OK;
}

int Engine::enqueue(const vt_frame* frames, int n, vt_result* host_res, StreamState* host_st) {
    if (!frames || n != B) return set_err(VT_ERR_INVALID_ARG, "enqueue: need exactly %d frames", B);
    for (int b = 0; b < B; ++b) {
        if (!h_initialized[b]) return set_err(VT_ERR_NOT_INITIALIZED, "stream %d: update before init", b);
        if (int rc = check_frame(frames[b])) return rc;
    }
    DEVICE_SCOPE(device);
    const int slot = ring_pos;
    ring_pos = (ring_pos + 1) % RING;
    HIPCHK(hipEventSynchronize(ring_ev[slot]));  // the copy that last used this slot is done
    FrameDesc* hf = h_block(slot);
    for (int b = 0; b < B; ++b) to_desc(frames[b], hf + b);
    *(PassOut*)(hf + B) = PassOut{host_res ? host_res : h_results, host_st ? host_st : h_states_all};
    HIPCHK(hipMemcpyAsync(d_frames, hf, frames_block_bytes(), hipMemcpyHostToDevice, stream));
    HIPCHK(hipEventRecord(ring_ev[slot], stream));
    const int tier = pick_crop_tier();
    if (use_graph && !taps) {
        if (!graph_exec[tier])      // not reached after a successful creation (capture_all_graphs); kept as the safe path
            if (int rc = capture_graph(tier)) return rc;
        HIPCHK(hipGraphLaunch(graph_exec[tier], stream));
        graph_replays[tier] += 1;
        return VT_OK;
    }
    crop_tier = tier;
    return run_pass(nullptr);
}

int Engine::wait(vt_result* out, int n) {
    if (n > B) n = B;
    DEVICE_SCOPE(device);
    HIPCHK(hipStreamSynchronize(stream));
    if (out)
        for (int b = 0; b < n; ++b) out[b] = h_results[b];
    if (host_seq == host_collected)                 // no pipelined pass owns the stream states
        for (int b = 0; b < B; ++b) known[b] = h_states_all[b];
    return VT_OK;
}

// ---- construction ----------------------------------------------------------------------------------

size_t nv12_bytes_read(size_t w, size_t h) {      // bytes the full-frame converter reads of a packed NV12 buffer
    if (!w || !h) return 0;
    const size_t uv_rows = (h + 1) / 2;
    const size_t last = (uv_rows - 1) * w + ((w & 1) ? w : w - 1);
    return w * h + last + 1;
}

int read_file(const char* path, std::vector<uint8_t>* out) {
    FILE* f = fopen(path, "rb");
    if (!f) return set_err(VT_ERR_IO, "cannot open weights file '%s'", path);
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0) { fclose(f); return set_err(VT_ERR_IO, "weights file '%s' is empty", path); }
    out->resize((size_t)sz);
    size_t got = fread(out->data(), 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) return set_err(VT_ERR_IO, "short read on '%s'", path);
    return VT_OK;
}

int check_device(int device_id) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return set_err(VT_ERR_NO_DEVICE, "no HIP device visible (%s); this library has no CPU path",
                       e == hipSuccess ? "count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= n)
        return set_err(VT_ERR_NO_DEVICE, "device %d out of range (have %d)", device_id, n);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) != hipSuccess)
        return set_err(VT_ERR_NO_DEVICE, "cannot query device %d", device_id);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return set_err(VT_ERR_NO_DEVICE, "device %d is %s; kernels are built for gfx950 only",
                       device_id, prop.gcnArchName);
    return VT_OK;
}

int make_engine(const char* path, const void* d_src, size_t bytes, int device_id,
                       const vt_config* cfg, int B, Engine** out) {
    if (!out) return set_err(VT_ERR_INVALID_ARG, "null output handle");
    *out = nullptr;
    if (B < 1 || B > VT_MAX_STREAMS)
        return set_err(VT_ERR_INVALID_ARG, "n_streams %d out of range (1..%d)", B, VT_MAX_STREAMS);
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    HIPCHK(gemm_prepare()); HIPCHK(attention_prepare()); HIPCHK(headconv_prepare());
    Engine* e = new (std::nothrow) Engine();
    if (!e) return set_err(VT_ERR_OOM, "out of host memory");
    e->device = device_id;
    e->B = B;
    int rc = VT_OK;
    do {
        hipError_t he = hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking);
        if (he != hipSuccess) { rc = set_err(VT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(he)); break; }
        if (path) {
            std::vector<uint8_t> blob;
            if ((rc = read_file(path, &blob))) break;
            if ((rc = e->load_blob_host(blob))) break;
        } else {
            if ((rc = e->load_blob_device(d_src, bytes))) break;
        }
        e->success_threshold = e->d.success_threshold;
        if (cfg && cfg->struct_size >= sizeof(vt_config)) {
            if (cfg->success_threshold >= 0.0f) e->success_threshold = cfg->success_threshold;
            e->use_graph = cfg->use_graph != 0;
            if (cfg->max_frame_width > 0) e->max_w = cfg->max_frame_width;
            if (cfg->max_frame_height > 0) e->max_h = cfg->max_frame_height;
            if (cfg->max_device_mib > 0) e->max_device_bytes = (size_t)cfg->max_device_mib << 20;
            if (cfg->host_window_margin_pct > 0) e->margin = std::min(cfg->host_window_margin_pct, 400) / 100.0f;
            else if (cfg->host_window_margin_pct < 0) e->margin = 0.0f;
            e->host_zero_copy = cfg->host_zero_copy;
        }
        if ((rc = e->alloc_buffers())) break;
        // which head: the band kernel only where the planner finds a band height for all three layer kinds
        e->head_band_ok = headconv_plannable(e->d.gs, e->d.C, e->d.C, e->d.D, false, false) &&
                          headconv_plannable(e->d.gs, e->d.C, e->d.C, 9 * e->d.C, true, false) &&
                          headconv_plannable(e->d.gs, e->d.C, e->d.C, 9 * e->d.C, true, true);
        if ((rc = e->capture_all_graphs())) break;
    } while (0);
    if (rc != VT_OK) {
        char keep[512];
        memcpy(keep, g_err, sizeof(keep));
        delete e;
        memcpy(g_err, keep, sizeof(keep));
        return rc;
    }
    *out = e;
    return VT_OK;
}

void fill_info(const Engine* e, vt_model_info* o) {
    memset(o, 0, sizeof(*o));
    const ModelDims& d = e->d;
    o->patch = d.patch; o->template_size = d.T; o->search_size = d.S; o->dim = d.D;
    o->heads = d.H; o->layers = d.L; o->mlp_dim = d.mlp; o->head_channels = d.C;
    o->tokens_template = d.nt; o->tokens_search = d.ns; o->kpad = d.kpad; o->score_grid = d.gs;
    o->encoder_flops_per_frame = e->flops_encoder();
    o->flops_per_frame = e->flops_encoder() + e->flops_head();
    o->weight_bytes = e->blob_bytes;
}
