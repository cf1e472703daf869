// vt_engine.hip — host side of libvittrack_hip.so: weight blob, per-GPU buffers, the per-frame
// launch plan (eager or one hipGraph replay), and the extern "C" ABI of include/vittrack_hip.h.
//
// One Engine = B independent tracked streams on one GPU. Everything a frame needs stays in HBM:
// the decode kernel of frame t writes the box that the preprocessing kernel of frame t+1 reads, so
// a stream of updates is a pure device-side chain; the host only supplies frame pointers and
// collects 24 B of result per stream.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <atomic>
#include <new>
#include <string>
#include <vector>

#include "vt_common.hpp"
#include <dlfcn.h>
#include <unistd.h>

// ---- error plumbing ------------------------------------------------------------------------------

static thread_local char g_err[512] = "";
static int set_err(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(expr)                                                                       \
    do {                                                                                   \
        hipError_t _e = (expr);                                                            \
        if (_e != hipSuccess)                                                              \
            return set_err(_e == hipErrorOutOfMemory ? VT_ERR_OOM : VT_ERR_HIP,            \
                           "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                           __LINE__);                                                      \
    } while (0)

// Nothing may unwind across the C boundary (the reference host is built with panic = "abort",
// /root/reference/Cargo.toml:37): every extern "C" entry is a function-try-block ending in one of
// these handlers. std::bad_alloc (vector / map / string / new inside the engine) -> VT_ERR_OOM.
#define VT_NOTHROW_INT                                                                      \
    catch (const std::bad_alloc&) { return set_err(VT_ERR_OOM, "out of host memory"); }     \
    catch (const std::exception& ex_) { return set_err(VT_ERR_HIP, "internal error: %s", ex_.what()); } \
    catch (...) { return set_err(VT_ERR_HIP, "internal error (unknown exception)"); }
#define VT_NOTHROW_VOID catch (...) { (void)set_err(VT_ERR_HIP, "internal error in a void entry point"); }
#define VT_NOTHROW_PTR catch (...) { (void)set_err(VT_ERR_HIP, "internal error"); return nullptr; }

// hipSetDevice for the duration of a call, restoring the caller's current device afterwards (a
// single-process multi-GPU host - or torch in the tests - keeps its own notion of "current").
struct DeviceScope {
    int prev = -1;
    hipError_t err = hipSuccess;
    explicit DeviceScope(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) err = hipSetDevice(dev); else prev = -1;
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
    DeviceScope(const DeviceScope&) = delete;
    DeviceScope& operator=(const DeviceScope&) = delete;
};
#define DEVICE_SCOPE(dev)                                                                   \
    DeviceScope dev_scope_(dev);                                                            \
    if (dev_scope_.err != hipSuccess)                                                       \
        return set_err(VT_ERR_HIP, "hipSetDevice(%d): %s", (int)(dev), hipGetErrorString(dev_scope_.err))

// ---- weight blob ---------------------------------------------------------------------------------

static const char kMagic[8] = {'V', 'T', 'W', 'B', '0', '0', '0', '1'};
static const size_t kHeaderBytes = 256, kEntryBytes = 64;

struct BlobEntry {
    char name[32];
    uint32_t dtype, rows, cols, pad;
    uint64_t offset, nbytes;
};
static_assert(sizeof(BlobEntry) == 64, "blob entry layout");

struct TensorRef {
    const void* ptr = nullptr;
    uint32_t dtype = 0, rows = 0, cols = 0;
};

struct LayerW {
    const float *ln1_g, *ln1_b, *qkv_b, *proj_b, *ln2_g, *ln2_b, *fc1_b, *fc2_b;
    const bf16_t *qkv_w, *proj_w, *fc1_w, *fc2_w;
    // LayerNorm 1 / 2 folded into the QKV / fc1 GEMM (launch_fold_layernorm, once per engine):
    // weights bf16(gamma * W), their column sums, and beta W^T + bias
    const bf16_t *qkv_wf = nullptr, *fc1_wf = nullptr;
    const float *qkv_cs = nullptr, *qkv_c = nullptr, *fc1_cs = nullptr, *fc1_c = nullptr;
};

struct KernelStat {
    std::string name;
    int launches = 0;
    double ms = 0, flops = 0, bytes = 0;
};

thread_local LaunchProbe* vt_launch_probe = nullptr;

struct Profiler {
    // a, b: marker events around the entry's launches; k: the first launch's own begin / end (vt_launch), used when the
    // entry was exactly one launch
    struct Rec { hipEvent_t a, b; LaunchProbe k; int fam; };
    std::vector<Rec> recs;
    std::vector<KernelStat> fams;
    int family(const std::string& n) {
        for (size_t i = 0; i < fams.size(); ++i)
            if (fams[i].name == n) return (int)i;
        KernelStat k;
        k.name = n;
        fams.push_back(k);
        return (int)fams.size() - 1;
    }
};


// kernel-family label of a GEMM launch (built only when a profiler is attached): the tile configuration
// in it is the one launch_gemm() really runs for these arguments
static const char* gemm_name(int epi, const GemmArgs& a) {
    static const char* tags[] = {"xpos", "xresid", "gelu", "relu", "qkv", "x"};
    static thread_local char buf[96];
    snprintf(buf, sizeof(buf), "gemm_bf16_%s_%s_n%dk%d", tags[epi], gemm_config_name(gemm_effective_config(a, epi)), a.N, a.K);
    return buf;
}

struct Engine {
    int device = 0, B = 1;
    bool use_graph = true, taps = false;
    ModelDims d{};
    hipStream_t stream = nullptr;
    // weights
    uint8_t* d_blob = nullptr;
    size_t blob_bytes = 0;
    std::map<std::string, TensorRef> tens;
    std::vector<LayerW> layers;
    // activations
    bf16_t *d_patches = nullptr, *d_qk = nullptr, *d_vt = nullptr,
           *d_attn = nullptr, *d_mlp = nullptr, *d_feat = nullptr, *d_ta = nullptr,
           *d_tb = nullptr, *d_zeros = nullptr;     // d_zeros: 256 B of zeros (out-of-map taps of the 3x3 convs)
    // residual stream as a bf16 pair (x = xh + xl), its chunk partial statistics and the row terms of the
    // folded LayerNorm (vt_common.hpp); folded weights of all layers
    bf16_t *d_xh = nullptr, *d_xl = nullptr, *d_foldw = nullptr, *d_taps = nullptr;
    unsigned* d_band_cnt = nullptr;               // per stream: bands of the last head layer that have arrived (k_head.hip)
    float* d_band_best = nullptr;                 // per stream and band: the band's argmax candidate
    float2 *d_cstat = nullptr, *d_rstat = nullptr;
    unsigned* d_panel_cnt = nullptr;   // arrival counters of the 256-row panels (X-epilogues of the 256x256 kernel)
    float *d_foldv = nullptr, *d_headout = nullptr;
    StreamState* d_states = nullptr;
    FrameDesc* d_frames = nullptr;
    vt_result* d_results = nullptr;
    // pinned host
    static const int RING = 8;
    FrameDesc* h_frames = nullptr;  // [RING] blocks of B descriptors + PassOut
    size_t frames_block_bytes() const { return sizeof(FrameDesc) * (size_t)B + sizeof(PassOut); }
    FrameDesc* h_block(int slot) const { return (FrameDesc*)((char*)h_frames + (size_t)slot * frames_block_bytes()); }
    hipEvent_t ring_ev[RING]{};
    int ring_pos = 0;
    vt_result* h_results = nullptr;
    StreamState* h_state = nullptr;
    // graph
    // one captured pass per crop-buffer tier (k_preproc.hip: 16 / 32 / 64 KiB of LDS per tile), captured when first needed
    static constexpr int TIERS = 3;
    hipGraph_t graph[TIERS] = {nullptr, nullptr, nullptr};
    hipGraphExec_t graph_exec[TIERS] = {nullptr, nullptr, nullptr};
    int crop_tier = 0;                            // tier of the pass being enqueued (from the boxes the host knows)
    int crop_tier_forced = -1;                    // >= 0: tests / A-B runs (vt_group_set_tuning "crop_tier")
    // host-pointer staging (single-stream API)
    uint8_t* d_stage = nullptr;
    uint8_t* h_pack = nullptr;      // pinned: the window of a host frame, packed
    size_t stage_bytes = 0;
    StreamState* h_states_all = nullptr;  // pinned mirror of d_states after the last pass
    int max_w = 3840, max_h = 2160;
    size_t max_device_bytes = 0;    // vt_config.max_device_mib (0: no limit but free memory)
    // host-side copy of the stream states after the last pass the HOST has collected (window planning
    // reads this, never a pinned buffer a running pass may still write)
    std::vector<StreamState> known;
    // pipelined host passes (vt_group_enqueue_host / vt_group_wait_next): two slots, each with its own
    // pinned + device arena, result buffers, state snapshot and events; uploads go on copy_stream
    struct HostSlot {
        uint8_t *d_arena = nullptr, *h_arena = nullptr;
        size_t bytes = 0;
        vt_result* h_res = nullptr;
        StreamState* h_st = nullptr;
        hipEvent_t up_ev = nullptr, done_ev = nullptr;
        std::vector<vt_frame> host;     // the caller's frames, valid until the pass is collected
        bool pending = false, speculative = false, redone = false;
    } hs[2];
    hipStream_t copy_stream = nullptr;
    unsigned host_seq = 0, host_collected = 0;   // pipelined passes enqueued / collected
    unsigned host_redos = 0;                      // passes redone because a speculative window missed
    float margin = 0.75f;                         // speculative enlargement of the crop side
    int head_band_kernel = 2;                     // 0: the head as implicit GEMMs + head_out + decode (A/B, tests);
                                                  // 1: band kernels behind the LayerNorm kernel; 2: + the final LayerNorm
                                                  // inside the 1x1 layer's kernel where the shape allows it (default)
    bool feat_in_head = false;                    // the passes do not write d_feat (recomputed when read)
    hipError_t final_layernorm();
    int host_zero_copy = 0;                       // vt_config.host_zero_copy: 0 auto (single-stream engines), 1 always, -1 never
    float success_threshold = 0.2f;
    std::vector<int> h_initialized;

    ~Engine() { destroy(); }
    void destroy();
    int load_blob_host(const std::vector<uint8_t>& blob);
    int load_blob_device(const void* d_src, size_t bytes);
    int index_blob(const uint8_t* host_copy, size_t bytes);
    size_t activation_bytes() const;
    int alloc_buffers();
    int run_pass(Profiler* prof);
    int capture_graph(int tier);
    int pick_crop_tier() const;
    void drop_graphs();
    // host_res / host_st: pinned buffers the pass's results and states are stored to (null: the
    // engine's own h_results / h_states_all)
    int enqueue(const vt_frame* frames, int n, vt_result* host_res = nullptr, StreamState* host_st = nullptr);
    int wait(vt_result* out, int n);
    int init_stream(int b, const vt_frame* f, vt_bbox box);
    const TensorRef* find(const std::string& n) const {
        auto it = tens.find(n);
        return it == tens.end() ? nullptr : &it->second;
    }
    double flops_encoder() const;
    double flops_head() const;
};

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

// zero-filled device buffer. The fill is ordered on the ENGINE's stream: that stream is non-blocking, so a
// hipMemset on the null stream (asynchronous for device memory) is not ordered against the kernels the
// engine launches next - the LayerNorm fold at construction raced with the fill of its own output when
// the null stream was busy zeroing the gigabytes of a several-hundred-stream engine.
template <typename T>
static hipError_t dalloc0(T** p, size_t count, hipStream_t s) {
    hipError_t e = hipMalloc((void**)p, count * sizeof(T));
    if (e != hipSuccess) return e;
    return hipMemsetAsync(*p, 0, count * sizeof(T), s);
}

// HBM the activations of B streams need (bytes), as alloc_buffers() lays them out
size_t Engine::activation_bytes() const {
    const size_t M = (size_t)B * d.ntok, Ms = (size_t)B * d.ns;
    const size_t fold_rows = (size_t)d.L * (3 * d.D + d.mlp);      // folded QKV + fc1 weights of every layer
    return 2 * (M * d.kpad + 2 * M * d.D + M * 2 * d.D + (size_t)B * d.H * 64 * d.npad + M * d.D + M * d.mlp +
                Ms * d.D + 2 * Ms * d.C + fold_rows * d.D) +
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
    HIPCHK(dalloc0(&d_xl, M * d.D, stream));
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
        // algorithmic bytes: operands once, output once; the residual pair is read and written (4 + 4 B)
        const double by = 2.0 * ((double)a.M * a.K + (double)a.N * a.K) +
                          (epi == EPI_RESID ? 8.0 : epi == EPI_F32_POS ? 4.0 : 2.0) * a.M * a.N;
        L(prof ? gemm_name(epi, a) : "", fl, by, [&] { return launch_gemm(a, epi, stream); });
    };
    auto tap = [&](int slot) {
        if (taps && lerr == hipSuccess) {     // both halves of the residual stream: [slot][hi | lo][M][D]
            bf16_t* dst = d_taps + (size_t)slot * 2 * M * D;
            lerr = hipMemcpyAsync(dst, d_xh, sizeof(bf16_t) * M * D, hipMemcpyDeviceToDevice, stream);
            if (lerr == hipSuccess)
                lerr = hipMemcpyAsync(dst + (size_t)M * D, d_xl, sizeof(bf16_t) * M * D, hipMemcpyDeviceToDevice, stream);
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
    // K2: patch embedding (+bias +pos) -> residual stream (bf16 pair + chunk statistics)
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
    const bool band = head_band_kernel && headconv_supported(d.gs, d.C, d.C, 9 * d.C, true) &&
                      headconv_supported(d.gs, d.C, d.C, D, false);
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
    graph[tier] = g;
    HIPCHK(hipGraphInstantiate(&graph_exec[tier], graph[tier], nullptr, nullptr, 0));
    return VT_OK;
}

static int check_frame(const vt_frame& f) {
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

static void to_desc(const vt_frame& f, FrameDesc* o) {
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
        if (!graph_exec[tier])
            if (int rc = capture_graph(tier)) return rc;
        HIPCHK(hipGraphLaunch(graph_exec[tier], stream));
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

static int read_file(const char* path, std::vector<uint8_t>* out) {
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

static int check_device(int device_id) {
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

static int make_engine(const char* path, const void* d_src, size_t bytes, int device_id,
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

static void fill_info(const Engine* e, vt_model_info* o) {
    memset(o, 0, sizeof(*o));
    const ModelDims& d = e->d;
    o->patch = d.patch; o->template_size = d.T; o->search_size = d.S; o->dim = d.D;
    o->heads = d.H; o->layers = d.L; o->mlp_dim = d.mlp; o->head_channels = d.C;
    o->tokens_template = d.nt; o->tokens_search = d.ns; o->kpad = d.kpad; o->score_grid = d.gs;
    o->encoder_flops_per_frame = e->flops_encoder();
    o->flops_per_frame = e->flops_encoder() + e->flops_head();
    o->weight_bytes = e->blob_bytes;
}

// ---- C ABI ---------------------------------------------------------------------------------------

struct vt_group { Engine* e; };
struct vt_tracker { Engine* e; vt_group view; };   // view: the tracker as a group of one

extern "C" {

void vt_config_default(vt_config* cfg) try {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(vt_config);
    cfg->success_threshold = -1.0f;
    cfg->use_graph = 1;
    cfg->n_streams = 1;
} VT_NOTHROW_VOID
const char* vt_last_error(void) { return g_err; }
int vt_abi_version(void) { return VT_ABI_VERSION; }
int vt_device_count(void) try {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    int ok = 0;
    for (int i = 0; i < n; ++i) {
        hipDeviceProp_t p;
        if (hipGetDeviceProperties(&p, i) == hipSuccess && strncmp(p.gcnArchName, "gfx950", 6) == 0) ++ok;
    }
    return ok;
} VT_NOTHROW_INT

struct vt_extmem {
    int device;
    hipExternalMemory_t mem;
};

int vt_import_dmabuf(int device_id, int fd, size_t bytes, vt_extmem** out, void** d_ptr) try {
    if (!out || !d_ptr || fd < 0 || bytes == 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    *out = nullptr; *d_ptr = nullptr;
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    const int dupfd = dup(fd);     // the import takes the descriptor over; the caller keeps its own
    if (dupfd < 0) return set_err(VT_ERR_INVALID_ARG, "dup(fd) failed");
    hipExternalMemoryHandleDesc hd;
    memset(&hd, 0, sizeof(hd));
    hd.type = hipExternalMemoryHandleTypeOpaqueFd;
    hd.handle.fd = dupfd;
    hd.size = bytes;
    hipExternalMemory_t mem = nullptr;
    hipError_t he = hipImportExternalMemory(&mem, &hd);
    if (he != hipSuccess) {
        close(dupfd);
        return set_err(VT_ERR_HIP, "hipImportExternalMemory(dma-buf): %s", hipGetErrorString(he));
    }
    hipExternalMemoryBufferDesc bd;
    memset(&bd, 0, sizeof(bd));
    bd.offset = 0; bd.size = bytes;
    void* p = nullptr;
    he = hipExternalMemoryGetMappedBuffer(&p, mem, &bd);
    if (he != hipSuccess || !p) {
        (void)hipDestroyExternalMemory(mem);
        return set_err(VT_ERR_HIP, "hipExternalMemoryGetMappedBuffer: %s", hipGetErrorString(he));
    }
    vt_extmem* xm = new (std::nothrow) vt_extmem{device_id, mem};
    if (!xm) { (void)hipDestroyExternalMemory(mem); return set_err(VT_ERR_OOM, "out of host memory"); }
    *out = xm;
    *d_ptr = p;
    return VT_OK;
} VT_NOTHROW_INT

void vt_release_dmabuf(vt_extmem* m) try {
    if (!m) return;
    DeviceScope ds(m->device);             // the caller's current device is restored on return
    (void)hipDeviceSynchronize();          // no kernel of ours may still read the mapping
    (void)hipDestroyExternalMemory(m->mem);
    delete m;
} VT_NOTHROW_VOID

int vt_export_dmabuf(int device_id, const void* d_ptr, size_t bytes, int* fd_out) try {
    if (!d_ptr || !fd_out || bytes == 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    *fd_out = -1;
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    int fd = -1;
    hipError_t he = hipMemGetHandleForAddressRange(&fd, (hipDeviceptr_t)d_ptr, bytes,
                                                   hipMemRangeHandleTypeDmaBufFd, 0);
    if (he != hipSuccess || fd < 0)
        return set_err(VT_ERR_HIP, "hipMemGetHandleForAddressRange(dma-buf): %s", hipGetErrorString(he));
    *fd_out = fd;
    return VT_OK;
} VT_NOTHROW_INT

// Host ranges mapped by vt_host_register: the host-pointer entry points look a frame's planes up here and, when
// both lie in a mapped range of the engine's device, hand the kernels the mapped pointers (no window packing, no
// staging copy). A handful of entries; a mutex, because registration and tracking run on different threads.
struct HostRange { const uint8_t* host; size_t bytes; uint8_t* dev; int device; };
static std::mutex g_ranges_mu;
static std::vector<HostRange> g_ranges;
static std::atomic<int> g_ranges_n{0};
// the whole extent [p, p + bytes) must lie inside one mapped range: a frame that only starts in one is staged
static const uint8_t* mapped_device_ptr(int device, const uint8_t* p, size_t bytes) {
    if (!p || bytes == 0 || g_ranges_n.load(std::memory_order_acquire) == 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    for (const HostRange& r : g_ranges)
        if (r.device == device && p >= r.host && p < r.host + r.bytes && bytes <= (size_t)(r.host + r.bytes - p))
            return r.dev + (p - r.host);
    return nullptr;
}

int vt_host_register(int device_id, void* host_ptr, size_t bytes, void** d_ptr) try {
    if (!host_ptr || !d_ptr || bytes == 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    *d_ptr = nullptr;
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    hipError_t he = hipHostRegister(host_ptr, bytes, hipHostRegisterMapped);
    if (he != hipSuccess) return set_err(VT_ERR_HIP, "hipHostRegister(%zu bytes): %s", bytes, hipGetErrorString(he));
    void* dp = nullptr;
    he = hipHostGetDevicePointer(&dp, host_ptr, 0);
    if (he != hipSuccess || !dp) {
        (void)hipHostUnregister(host_ptr);
        return set_err(VT_ERR_HIP, "hipHostGetDevicePointer: %s", hipGetErrorString(he));
    }
    *d_ptr = dp;
    {
        std::lock_guard<std::mutex> lk(g_ranges_mu);
        g_ranges.push_back(HostRange{(const uint8_t*)host_ptr, bytes, (uint8_t*)dp, device_id});
        g_ranges_n.store((int)g_ranges.size(), std::memory_order_release);
    }
    return VT_OK;
} VT_NOTHROW_INT

int vt_host_unregister(int device_id, void* host_ptr) try {
    if (!host_ptr) return set_err(VT_ERR_INVALID_ARG, "null pointer");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    {
        std::lock_guard<std::mutex> lk(g_ranges_mu);
        for (size_t i = 0; i < g_ranges.size(); ++i)
            if (g_ranges[i].host == (const uint8_t*)host_ptr && g_ranges[i].device == device_id) {
                g_ranges.erase(g_ranges.begin() + (long)i);
                break;
            }
        g_ranges_n.store((int)g_ranges.size(), std::memory_order_release);
    }
    (void)hipDeviceSynchronize();          // no kernel of ours may still read the mapping
    hipError_t he = hipHostUnregister(host_ptr);
    if (he != hipSuccess) return set_err(VT_ERR_HIP, "hipHostUnregister: %s", hipGetErrorString(he));
    return VT_OK;
} VT_NOTHROW_INT

int vt_recommended_streams(const vt_model_info* info, int max_streams) try {
    if (!info || info->dim <= 0 || (info->dim % 256) != 0 || max_streams < 1) return 1;
    const long tokens = (long)info->tokens_template + info->tokens_search;
    const long cols[3] = {info->dim / 256, 3L * info->dim / 256, info->mlp_dim / 256};
    for (int b = 1; b <= max_streams; ++b) {
        const long rows = (b * tokens + 255) / 256;
        bool ok = true;
        for (long c : cols) {
            const long t = rows * c, rounds = (t + 255) / 256;
            if (c <= 0 || t * 100 < rounds * 256 * 98) ok = false;   // < 98 % of the rounds' CU slots used
        }
        if (ok) return b;
    }
    return 1;
} VT_NOTHROW_INT

// largest engine for which every encoder GEMM still runs on the 256x256 kernels: they address an operand
// with unsigned 32-bit byte offsets (k_gemm256.hip launch_gemm256), and the widest A operand of a pass is
// max(dim, mlp_dim, kpad) bf16 columns by B * tokens rows
static int engine_stream_cap(const vt_model_info* info) {
    const long long tokens = (long long)info->tokens_template + info->tokens_search;
    const long long width = std::max(std::max((long long)info->dim, (long long)info->mlp_dim), (long long)info->kpad);
    if (tokens <= 0 || width <= 0) return VT_MAX_STREAMS;
    const long long b = (VT_GEMM256_MAX_OPERAND_BYTES - 1) / (tokens * width * 2);
    return (int)std::max(1LL, std::min((long long)VT_MAX_STREAMS, b));
}

int vt_plan_engines(const vt_model_info* info, int n_streams, int* sizes, int cap) try {
    if (!info || !sizes || n_streams < 1 || cap < 1) return 0;
    const int r = vt_recommended_streams(info, 128);
    const int bmax = engine_stream_cap(info);
    int k;                                        // engines
    if (r <= 1 || n_streams <= r) k = 1;
    else k = 2;
    while ((n_streams + k - 1) / k > bmax) ++k;
    if (k > cap) return 0;
    if (k == 2 && n_streams < 2 * r) {            // a full engine and the rest
        sizes[0] = r;
        sizes[1] = n_streams - r;
        return 2;
    }
    for (int i = 0; i < k; ++i) sizes[i] = n_streams / k + (i < n_streams % k ? 1 : 0);
    return k;
} VT_NOTHROW_INT

int vt_group_create(const char* weights_path, int device_id, const vt_config* cfg, vt_group** out) try {
    if (!weights_path || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = nullptr;
    const int B = (cfg && cfg->struct_size >= sizeof(vt_config) && cfg->n_streams > 0) ? cfg->n_streams : 1;
    if (int rc = make_engine(weights_path, nullptr, 0, device_id, cfg, B, &e)) return rc;
    vt_group* g = new (std::nothrow) vt_group{e};
    if (!g) { delete e; return set_err(VT_ERR_OOM, "out of host memory"); }
    *out = g;
    return VT_OK;
} VT_NOTHROW_INT
int vt_group_create_from_device_blob(const void* d_blob, size_t bytes, int device_id,
                                     const vt_config* cfg, vt_group** out) try {
    if (!d_blob || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = nullptr;
    const int B = (cfg && cfg->struct_size >= sizeof(vt_config) && cfg->n_streams > 0) ? cfg->n_streams : 1;
    if (int rc = make_engine(nullptr, d_blob, bytes, device_id, cfg, B, &e)) return rc;
    vt_group* g = new (std::nothrow) vt_group{e};
    if (!g) { delete e; return set_err(VT_ERR_OOM, "out of host memory"); }
    *out = g;
    return VT_OK;
} VT_NOTHROW_INT
void vt_group_destroy(vt_group* g) try {
    if (!g) return;
    delete g->e;
    delete g;
} VT_NOTHROW_VOID
int vt_group_streams(const vt_group* g) { return g ? g->e->B : 0; }
int vt_group_get_model_info(const vt_group* g, vt_model_info* out) try {
    if (!g || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    fill_info(g->e, out);
    return VT_OK;
} VT_NOTHROW_INT
// A pipelined host pass (vt_group_enqueue_host) that has not been collected owns the stream states:
// its redo path rewinds to the host's copy of them (`known`). Everything that would advance or
// overwrite the states behind such a pass is refused until vt_group_wait_next has collected it.
static int refuse_while_pipelined(const Engine* e, const char* what) {
    if (e->host_seq != e->host_collected)
        return set_err(VT_ERR_INVALID_ARG, "%s: collect the pipelined host passes first (vt_group_wait_next)", what);
    return VT_OK;
}

int vt_group_init_device(vt_group* g, int stream, const vt_frame* frame, vt_bbox box) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    if (int rc = refuse_while_pipelined(g->e, "init")) return rc;
    return g->e->init_stream(stream, frame, box);
} VT_NOTHROW_INT
int vt_group_enqueue_device(vt_group* g, const vt_frame* frames, int n) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    if (int rc = refuse_while_pipelined(g->e, "enqueue_device")) return rc;
    return g->e->enqueue(frames, n);
} VT_NOTHROW_INT
int vt_group_wait(vt_group* g, vt_result* out, int n) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    if (int rc = refuse_while_pipelined(g->e, "wait")) return rc;      // its results live in the pass's slot
    return g->e->wait(out, n);
} VT_NOTHROW_INT
int vt_group_update_device(vt_group* g, const vt_frame* frames, int n, vt_result* out) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    if (int rc = refuse_while_pipelined(g->e, "update_device")) return rc;
    if (int rc = g->e->enqueue(frames, n)) return rc;
    return g->e->wait(out, n);
} VT_NOTHROW_INT
void* vt_group_hip_stream(vt_group* g) { return g ? (void*)g->e->stream : nullptr; }

static int stage_host_frames(Engine* e, const vt_frame* host, int n, const float (*boxes)[4], vt_frame* dev);

int vt_group_init_host(vt_group* g, int stream, const vt_frame* host_frame, vt_bbox box) try {
    if (!g || !host_frame) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = g->e;
    if (stream < 0 || stream >= e->B) return set_err(VT_ERR_INVALID_ARG, "bad stream index");
    if (int rc = refuse_while_pipelined(e, "init_host")) return rc;
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));     // the staging arena is shared by the group's passes
    const float fb[1][4] = {{(float)box.x, (float)box.y, (float)box.width, (float)box.height}};
    vt_frame f;
    if (int rc = stage_host_frames(e, host_frame, 1, fb, &f)) return rc;
    return e->init_stream(stream, &f, box);
} VT_NOTHROW_INT

int vt_group_update_host(vt_group* g, const vt_frame* host_frames, int n, vt_result* out) try {
    if (!g || !host_frames || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = g->e;
    if (n != e->B) return set_err(VT_ERR_INVALID_ARG, "update_host: need exactly %d frames", e->B);
    for (int b = 0; b < n; ++b)
        if (!e->h_initialized[b]) return set_err(VT_ERR_NOT_INITIALIZED, "stream %d not initialised", b);
    if (e->host_seq != e->host_collected)
        return set_err(VT_ERR_INVALID_ARG, "update_host: collect the pipelined passes first (vt_group_wait_next)");
    DEVICE_SCOPE(e->device);
    if (int rc = e->wait(nullptr, 0)) return rc;   // last pass done: its boxes are in `known`
    std::vector<vt_frame> dev((size_t)n);
    std::vector<float> boxes((size_t)n * 4);
    for (int b = 0; b < n; ++b) memcpy(&boxes[(size_t)b * 4], e->known[b].box, 4 * sizeof(float));
    if (int rc = stage_host_frames(e, host_frames, n, reinterpret_cast<const float(*)[4]>(boxes.data()), dev.data()))
        return rc;
    if (int rc = e->enqueue(dev.data(), n)) return rc;
    return e->wait(out, n);
} VT_NOTHROW_INT

int vt_group_enable_taps(vt_group* g, int enable) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    Engine* e = g->e;
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (enable && !e->d_taps)      // per slot: the hi and the lo half of the residual stream
        HIPCHK(dalloc0(&e->d_taps, (size_t)(e->d.L + 1) * 2 * e->B * e->d.ntok * e->d.D, e->stream));
    e->taps = enable != 0;
    return VT_OK;
} VT_NOTHROW_INT

int vt_group_set_tuning(vt_group* g, const char* key, int value) try {
    if (!g || !key) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = g->e;
    if (int rc = refuse_while_pipelined(e, "set_tuning")) return rc;
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    const std::string k = key;
    if (k == "head_band") e->head_band_kernel = value < 0 ? 2 : value;   // 0 / 1 / 2, see Engine::head_band_kernel
    else if (k == "crop_tier") e->crop_tier_forced = value;      // < 0: chosen per pass from the known boxes (default)
    else return set_err(VT_ERR_INVALID_ARG, "unknown tuning key '%s'", key);
    // the captured passes hold the old choice: drop them, the next pass captures again
    e->drop_graphs();
    return VT_OK;
} VT_NOTHROW_INT

int vt_group_set_state_box(vt_group* g, int stream, const float* box4) try {
    if (!g || !box4) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = g->e;
    if (stream < 0 || stream >= e->B) return set_err(VT_ERR_INVALID_ARG, "bad stream index");
    if (!e->h_initialized[stream]) return set_err(VT_ERR_NOT_INITIALIZED, "stream %d not initialised", stream);
    if (int rc = refuse_while_pipelined(e, "set_state_box")) return rc;
    for (int k = 0; k < 4; ++k)
        if (!std::isfinite(box4[k])) return set_err(VT_ERR_INVALID_ARG, "state box: non-finite value");
    if (!(box4[2] >= 1.0f) || !(box4[3] >= 1.0f) || box4[2] > 32768.0f || box4[3] > 32768.0f ||
        fabsf(box4[0]) > 65536.0f || fabsf(box4[1]) > 65536.0f)
        return set_err(VT_ERR_INVALID_ARG, "state box %g,%g %gx%g out of range", box4[0], box4[1],
                       box4[2], box4[3]);
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    HIPCHK(hipMemcpy(e->d_states[stream].box, box4, 4 * sizeof(float), hipMemcpyHostToDevice));
    memcpy(e->h_states_all[stream].box, box4, 4 * sizeof(float));
    memcpy(e->known[stream].box, box4, 4 * sizeof(float));
    return VT_OK;
} VT_NOTHROW_INT

int vt_group_profile_device(vt_group* g, const vt_frame* frames, int n, int iters,
                            vt_kernel_time* out, int max_out) try {
    if (!g || !frames || !out || iters < 1) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    Engine* e = g->e;
    if (n != e->B) return set_err(VT_ERR_INVALID_ARG, "profile: need exactly %d frames", e->B);
    if (int rc = refuse_while_pipelined(e, "profile")) return rc;
    for (int b = 0; b < e->B; ++b)
        if (!e->h_initialized[b]) return set_err(VT_ERR_NOT_INITIALIZED, "stream %d not initialised", b);
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    FrameDesc* hf = e->h_block(0);
    for (int b = 0; b < e->B; ++b) {
        if (int rc = check_frame(frames[b])) return rc;
        to_desc(frames[b], hf + b);
    }
    *(PassOut*)(hf + e->B) = PassOut{e->h_results, e->h_states_all};
    HIPCHK(hipMemcpyAsync(e->d_frames, hf, e->frames_block_bytes(), hipMemcpyHostToDevice, e->stream));
    Profiler prof;
    e->crop_tier = e->pick_crop_tier();
    for (int it = 0; it < iters; ++it)
        if (int rc = e->run_pass(&prof)) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    for (auto& r : prof.recs) {
        float ms = 0;
        // one launch: its own begin -> end; several (or none through vt_launch): the markers around them
        if (r.k.launches != 1 || hipEventElapsedTime(&ms, r.k.start, r.k.stop) != hipSuccess || ms <= 0.0f)
            (void)hipEventElapsedTime(&ms, r.a, r.b);
        prof.fams[r.fam].ms += ms;
        (void)hipEventDestroy(r.a);
        (void)hipEventDestroy(r.b);
        (void)hipEventDestroy(r.k.start);
        (void)hipEventDestroy(r.k.stop);
    }
    int k = 0;
    for (auto& f : prof.fams) {
        if (k >= max_out) break;
        vt_kernel_time& o = out[k++];
        memset(&o, 0, sizeof(o));
        snprintf(o.name, sizeof(o.name), "%s", f.name.c_str());
        o.launches = f.launches / iters;
        o.ms_total = (float)(f.ms / iters);
        o.flops = f.flops / iters;
        o.bytes = f.bytes / iters;
    }
    return k;
} VT_NOTHROW_INT

static int64_t copy_out_f32(const float* dsrc, int64_t count, float* out, int64_t cap) {
    if (!out) return count;
    if (cap < count) return set_err(VT_ERR_INVALID_ARG, "read_tensor: capacity %lld < %lld", (long long)cap, (long long)count);
    if (hipMemcpy(out, dsrc, sizeof(float) * count, hipMemcpyDeviceToHost) != hipSuccess)
        return set_err(VT_ERR_HIP, "read_tensor: copy failed");
    return count;
}
// the residual stream: hi + lo in float32 (the value the bf16 pair stands for)
static int64_t copy_out_pair(const bf16_t* dhi, const bf16_t* dlo, int64_t count, float* out, int64_t cap) {
    if (!out) return count;
    if (cap < count) return set_err(VT_ERR_INVALID_ARG, "read_tensor: capacity %lld < %lld", (long long)cap, (long long)count);
    std::vector<bf16_t> hi((size_t)count), lo((size_t)count);
    if (hipMemcpy(hi.data(), dhi, 2 * count, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(lo.data(), dlo, 2 * count, hipMemcpyDeviceToHost) != hipSuccess)
        return set_err(VT_ERR_HIP, "read_tensor: copy failed");
    for (int64_t i = 0; i < count; ++i) {
        const uint32_t uh = ((uint32_t)hi[i]) << 16, ul = ((uint32_t)lo[i]) << 16;
        float fh, fl;
        memcpy(&fh, &uh, 4); memcpy(&fl, &ul, 4);
        out[i] = fh + fl;
    }
    return count;
}
static int64_t copy_out_bf16(const bf16_t* dsrc, int64_t count, float* out, int64_t cap) {
    if (!out) return count;
    if (cap < count) return set_err(VT_ERR_INVALID_ARG, "read_tensor: capacity %lld < %lld", (long long)cap, (long long)count);
    std::vector<bf16_t> tmp((size_t)count);
    if (hipMemcpy(tmp.data(), dsrc, 2 * count, hipMemcpyDeviceToHost) != hipSuccess)
        return set_err(VT_ERR_HIP, "read_tensor: copy failed");
    for (int64_t i = 0; i < count; ++i) {
        uint32_t u = ((uint32_t)tmp[i]) << 16;
        memcpy(out + i, &u, 4);
    }
    return count;
}

int64_t vt_group_read_tensor(vt_group* g, int stream, const char* name, float* out, int64_t capacity) try {
    if (!g || !name) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = g->e;
    if (stream < 0 || stream >= e->B) return set_err(VT_ERR_INVALID_ARG, "bad stream index");
    DEVICE_SCOPE(e->device);
    if (hipStreamSynchronize(e->stream) != hipSuccess)
        return set_err(VT_ERR_HIP, "read_tensor: sync failed");
    const ModelDims& d = e->d;
    const std::string n(name);
    const size_t b = (size_t)stream;
    if (n == "patches") return copy_out_bf16(e->d_patches + b * d.ntok * d.kpad, (int64_t)d.ntok * d.kpad, out, capacity);
    if (n == "feat" && e->feat_in_head) {       // the pass normalised the rows inside the head's first kernel: same arithmetic, now
        HIPCHK(e->final_layernorm());        // as a launch (the residual stream of the last pass is still in place)
        HIPCHK(hipStreamSynchronize(e->stream));
    }
    if (n == "feat") return copy_out_bf16(e->d_feat + b * d.ns * d.D, (int64_t)d.ns * d.D, out, capacity);
    if (n == "attn") return copy_out_bf16(e->d_attn + b * d.ntok * d.D, (int64_t)d.ntok * d.D, out, capacity);
    if (n == "head_t3") return copy_out_bf16(e->d_tb + b * d.ns * d.C, (int64_t)d.ns * d.C, out, capacity);
    if (n == "head_out") return copy_out_f32(e->d_headout + b * d.ns * 8, (int64_t)d.ns * 8, out, capacity);
    if (n == "x") return copy_out_pair(e->d_xh + b * d.ntok * d.D, e->d_xl + b * d.ntok * d.D, (int64_t)d.ntok * d.D, out, capacity);
    if (n == "rowstat") return copy_out_f32((const float*)(e->d_rstat + b * d.ntok), (int64_t)d.ntok * 2, out, capacity);
    if (n == "state") {
        static_assert(sizeof(StreamState) % 4 == 0, "state size");
        return copy_out_f32((const float*)(e->d_states + b), sizeof(StreamState) / 4, out, capacity);
    }
    int slot = -1;
    if (n == "tokens0") slot = 0;
    else if (n.rfind("layer", 0) == 0) slot = 1 + atoi(n.c_str() + 5);
    if (slot >= 0 && slot <= d.L) {
        if (!e->d_taps) return set_err(VT_ERR_INVALID_ARG, "taps not enabled (vt_group_enable_taps)");
        const size_t M = (size_t)e->B * d.ntok;
        const bf16_t* hi = e->d_taps + ((size_t)slot * 2 * M + b * d.ntok) * d.D;
        return copy_out_pair(hi, hi + M * d.D, (int64_t)d.ntok * d.D, out, capacity);
    }
    return set_err(VT_ERR_INVALID_ARG, "unknown tensor '%s'", name);
} VT_NOTHROW_INT

// ---- single-stream drop-in --------------------------------------------------------------------------

int vt_create(const char* weights_path, int device_id, const vt_config* cfg, vt_tracker** out) try {
    if (!weights_path || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = nullptr;
    if (int rc = make_engine(weights_path, nullptr, 0, device_id, cfg, 1, &e)) return rc;
    vt_tracker* t = new (std::nothrow) vt_tracker{e, {e}};
    if (!t) { delete e; return set_err(VT_ERR_OOM, "out of host memory"); }
    *out = t;
    return VT_OK;
} VT_NOTHROW_INT
int vt_create_from_device_blob(const void* d_blob, size_t bytes, int device_id, const vt_config* cfg,
                               vt_tracker** out) try {
    if (!d_blob || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = nullptr;
    if (int rc = make_engine(nullptr, d_blob, bytes, device_id, cfg, 1, &e)) return rc;
    vt_tracker* t = new (std::nothrow) vt_tracker{e, {e}};
    if (!t) { delete e; return set_err(VT_ERR_OOM, "out of host memory"); }
    *out = t;
    return VT_OK;
} VT_NOTHROW_INT
void vt_destroy(vt_tracker* t) try {
    if (!t) return;
    delete t->e;
    delete t;
} VT_NOTHROW_VOID
int vt_get_model_info(const vt_tracker* t, vt_model_info* out) try {
    if (!t || !out) return set_err(VT_ERR_INVALID_ARG, "null argument");
    fill_info(t->e, out);
    return VT_OK;
} VT_NOTHROW_INT

// Host-pointer ingest: only the window of the frame that the call can sample is uploaded. The
// reference hands over the whole frame (6.2 MB of RGB8 at 1080p, src/pipeline.rs:105-112) although
// the tracker reads a window of side 4*sqrt(w*h) around the last box; that window is packed into a
// pinned buffer on the host (a few hundred KB) and copied asynchronously ahead of the kernels.
// The caller's buffer is no longer referenced when this returns (src/pipeline.rs:125 draws into it).
// ---- host-frame ingest: only the windows that the pass can sample cross PCIe ----------------------
// The reference hands over whole frames (6.2 MB of RGB8 at 1080p, src/pipeline.rs:105-112) although
// the tracker reads a window of side 4*sqrt(w*h) around the last box. The windows of all the frames
// of a call are packed back to back into one pinned arena and moved with ONE async H2D copy; the
// frame descriptors handed to the kernels point into the device copy and carry the window origin.
struct HostWin {
    int fmt, w, h, s0, s1;
    const uint8_t *p0, *p1;
    int x_lo, y_lo, ww, wh;
    size_t bytes, uv_off;
};

// `grow`: enlargement of the crop side for a SPECULATIVE window (the box of the pass that is still
// running is not known): 0 = the exact crop
static int plan_window(const Engine* e, int fmt, const uint8_t* p0, const uint8_t* p1, int w, int h,
                       int s0, int s1, const float* box, float grow, HostWin* win) {
    if (!p0 || w < 16 || h < 16) return set_err(VT_ERR_INVALID_ARG, "null frame or size < 16");
    if (w > e->max_w || h > e->max_h)
        return set_err(VT_ERR_INVALID_ARG, "frame %dx%d exceeds configured max %dx%d", w, h, e->max_w, e->max_h);
    if (fmt == VT_PIX_RGB8) {
        if (s0 < 3 * w) return set_err(VT_ERR_INVALID_ARG, "rgb8 stride < 3*width");
    } else if (fmt == VT_PIX_YUY2) {
        if ((w & 1) || s0 < 2 * w) return set_err(VT_ERR_INVALID_ARG, "yuy2: odd width or stride < 2*width");
    } else if (fmt == VT_PIX_NV12) {
        if (!p1 || s0 < w || s1 < ((w + 1) & ~1)) return set_err(VT_ERR_INVALID_ARG, "nv12: bad plane or stride");
    } else {
        return set_err(VT_ERR_INVALID_ARG, "unknown pixel format %d", fmt);
    }
    // window = search crop (factor 4; it contains the factor-2 template crop) + bilinear margin
    const float side = 4.0f * sqrtf(fmaxf(box[2] * box[3], 1.0f)) * (1.0f + grow);
    const float cx = box[0] + 0.5f * box[2], cy = box[1] + 0.5f * box[3];
    long x_lo = (long)floorf(cx - 0.5f * side) - 4, x_hi = (long)ceilf(cx + 0.5f * side) + 4;
    long y_lo = (long)floorf(cy - 0.5f * side) - 4, y_hi = (long)ceilf(cy + 0.5f * side) + 4;
    x_lo = std::max(0L, std::min((long)w, x_lo)) & ~1L;
    y_lo = std::max(0L, std::min((long)h, y_lo)) & ~1L;
    x_hi = std::max(x_lo, std::min((long)w, (x_hi + 1) & ~1L));
    y_hi = std::max(y_lo, std::min((long)h, (y_hi + 1) & ~1L));
    if (x_hi - x_lo < 2 || y_hi - y_lo < 2) {   // window misses the frame: nothing can be sampled
        x_lo = 0; y_lo = 0; x_hi = 2; y_hi = 2;
    }
    win->fmt = fmt; win->w = w; win->h = h; win->s0 = s0; win->s1 = s1; win->p0 = p0; win->p1 = p1;
    win->x_lo = (int)x_lo; win->y_lo = (int)y_lo;
    win->ww = (int)(x_hi - x_lo); win->wh = (int)(y_hi - y_lo);
    if (fmt == VT_PIX_NV12) {
        // rows of the packed window start on 16-byte boundaries: the pixel kernel then fetches 8 pixels per load
        const int uvh = (win->wh + 1) / 2;
        const size_t ys = ((size_t)win->ww + 15) & ~(size_t)15, uvs = ((size_t)((win->ww + 1) & ~1) + 15) & ~(size_t)15;
        win->uv_off = (ys * win->wh + 255) & ~(size_t)255;
        win->bytes = win->uv_off + uvs * uvh;
    } else {
        win->uv_off = 0;
        win->bytes = (size_t)win->ww * win->wh * (fmt == VT_PIX_RGB8 ? 3 : 2);
    }
    win->bytes = (win->bytes + 255) & ~(size_t)255;
    return VT_OK;
}

// where the packed windows of one call go: a pinned host arena, its device twin, and the stream the
// single H2D copy is enqueued on
struct Arena {
    uint8_t** d;
    uint8_t** h;
    size_t* cap;
    hipStream_t copy_on;
};

// pinned + device arena of at least `need` bytes (grown only while nothing uses it)
static int ensure_arena(Engine* e, const Arena& a, size_t need) {
    if (need <= *a.cap) return VT_OK;
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (a.copy_on != e->stream) HIPCHK(hipStreamSynchronize(a.copy_on));
    if (*a.d) { (void)hipFree(*a.d); *a.d = nullptr; }
    if (*a.h) { (void)hipHostFree(*a.h); *a.h = nullptr; }
    *a.cap = 0;
    const size_t cap = need + need / 2 + 4096;
    HIPCHK(hipMalloc((void**)a.d, cap));
    HIPCHK(hipHostMalloc((void**)a.h, cap));
    *a.cap = cap;
    return VT_OK;
}

static void pack_window(const Arena& a, const HostWin& wn, size_t off, vt_frame* f) {
    uint8_t* dst = *a.h + off;
    memset(f, 0, sizeof(*f));
    f->width = wn.w; f->height = wn.h; f->format = wn.fmt;
    f->origin_x = wn.x_lo; f->origin_y = wn.y_lo;
    f->windowed = 1;   // strides describe the packed window
    f->window_w = wn.ww; f->window_h = wn.wh;
    if (wn.fmt == VT_PIX_RGB8 || wn.fmt == VT_PIX_YUY2) {
        const size_t bpp = wn.fmt == VT_PIX_RGB8 ? 3 : 2;
        const size_t rb = (size_t)wn.ww * bpp;
        for (int r = 0; r < wn.wh; ++r)
            memcpy(dst + r * rb, wn.p0 + (size_t)(wn.y_lo + r) * wn.s0 + (size_t)wn.x_lo * bpp, rb);
        f->plane0 = *a.d + off; f->stride0 = (int)rb;
    } else {
        const int uvw = (wn.ww + 1) & ~1, uvh = (wn.wh + 1) / 2;
        const size_t ys = ((size_t)wn.ww + 15) & ~(size_t)15, uvs = ((size_t)uvw + 15) & ~(size_t)15;   // as plan_window
        for (int r = 0; r < wn.wh; ++r)
            memcpy(dst + (size_t)r * ys, wn.p0 + (size_t)(wn.y_lo + r) * wn.s0 + wn.x_lo, (size_t)wn.ww);
        // odd frame width: the last pixel's V byte lies one past the row's last full pair
        const int uv_avail = (int)std::min<long>(uvw, (long)wn.s1 - wn.x_lo);
        for (int r = 0; r < uvh; ++r)
            memcpy(dst + wn.uv_off + (size_t)r * uvs, wn.p1 + (size_t)(wn.y_lo / 2 + r) * wn.s1 + wn.x_lo,
                   (size_t)uv_avail);
        f->plane0 = *a.d + off; f->plane1 = *a.d + off + wn.uv_off;
        f->stride0 = (int)ys; f->stride1 = (int)uvs;
    }
}

// n host frames -> n device frame descriptors (windows packed, ONE H2D copy enqueued on a.copy_on).
// boxes[i]: the box that decides stream i's window (the new box at init, the last state at update).
static int stage_host_frames_to(Engine* e, const Arena& a, const vt_frame* host, int n, const float (*boxes)[4],
                                float grow, vt_frame* dev, size_t* bytes_out) {
    std::vector<HostWin> wins((size_t)n);
    std::vector<char> mapped((size_t)n, 0);
    size_t total = 0;
    for (int i = 0; i < n; ++i) {
        const vt_frame& hf = host[i];
        if (int rc = plan_window(e, hf.format, (const uint8_t*)hf.plane0, (const uint8_t*)hf.plane1, hf.width,
                                 hf.height, hf.stride0, hf.stride1, boxes[i], grow, &wins[i]))
            return rc;
        // a frame inside a range mapped by vt_host_register goes to the kernels as it lies (zero copy) - on
        // single-stream engines, or where the caller asked for it: for a batched engine the packed upload beside
        // the previous pass is faster than PCIe reads inside the pass (vt_config.host_zero_copy, vittrack_hip.h)
        const bool zc = e->host_zero_copy > 0 || (e->host_zero_copy == 0 && e->B == 1);
        if (!zc) { total += wins[i].bytes; continue; }
        // bytes the kernels may touch: every row of the frame, the last one only as far as it is wide
        const size_t rowb = hf.format == VT_PIX_NV12 ? (size_t)hf.width : hf.format == VT_PIX_RGB8 ? (size_t)hf.width * 3 : (size_t)hf.width * 2;
        const size_t ext0 = (size_t)(hf.height - 1) * (size_t)hf.stride0 + rowb;
        const size_t ext1 = hf.format == VT_PIX_NV12 ? (size_t)((hf.height + 1) / 2 - 1) * (size_t)hf.stride1 + (size_t)((hf.width + 1) & ~1) : 0;
        const uint8_t* d0 = mapped_device_ptr(e->device, (const uint8_t*)hf.plane0, ext0);
        const uint8_t* d1 = hf.format == VT_PIX_NV12 ? mapped_device_ptr(e->device, (const uint8_t*)hf.plane1, ext1) : nullptr;
        if (d0 && (hf.format != VT_PIX_NV12 || d1)) {
            mapped[(size_t)i] = 1;
            memset(&dev[i], 0, sizeof(vt_frame));
            dev[i].plane0 = d0; dev[i].plane1 = d1; dev[i].width = hf.width; dev[i].height = hf.height;
            dev[i].stride0 = hf.stride0; dev[i].stride1 = hf.stride1; dev[i].format = hf.format;
            continue;
        }
        total += wins[i].bytes;
    }
    if (bytes_out) *bytes_out = total;
    if (total == 0) return VT_OK;            // every frame mapped: nothing to pack, nothing to copy
    if (int rc = ensure_arena(e, a, total)) return rc;
    DEVICE_SCOPE(e->device);
    size_t off = 0;
    for (int i = 0; i < n; ++i) {
        if (mapped[(size_t)i]) continue;
        pack_window(a, wins[i], off, &dev[i]);
        off += wins[i].bytes;
    }
    HIPCHK(hipMemcpyAsync(*a.d, *a.h, total, hipMemcpyHostToDevice, a.copy_on));
    if (bytes_out) *bytes_out = total;
    return VT_OK;
}

// the synchronous entry points: one arena, copy on the engine's own stream (every such call waits
// for its pass before returning, so the arena is free again at the next call)
static int stage_host_frames(Engine* e, const vt_frame* host, int n, const float (*boxes)[4], vt_frame* dev) {
    const Arena a{&e->d_stage, &e->h_pack, &e->stage_bytes, e->stream};
    return stage_host_frames_to(e, a, host, n, boxes, 0.0f, dev, nullptr);
}

static int stage_host_frame(Engine* e, int fmt, const uint8_t* p0, const uint8_t* p1, int w, int h,
                            int s0, int s1, const float* box, vt_frame* f) {
    vt_frame hf;
    memset(&hf, 0, sizeof(hf));
    hf.plane0 = p0; hf.plane1 = p1; hf.width = w; hf.height = h; hf.stride0 = s0; hf.stride1 = s1; hf.format = fmt;
    float b4[1][4] = {{box[0], box[1], box[2], box[3]}};
    return stage_host_frames(e, &hf, 1, b4, f);
}

// ---- pipelined host passes -------------------------------------------------------------------------

static int host_slot_prepare(Engine* e, Engine::HostSlot& sl) {
    if (sl.h_res) return VT_OK;
    DEVICE_SCOPE(e->device);
    // HIP multiplexes a process's streams onto a few hardware queues (four by default): with more
    // streams than that alive - e.g. four engines, each with a compute and a copy stream - an upload
    // can share a queue with some engine's compute stream and is then ordered behind that engine's
    // whole pass (measured: pipelined = synchronous throughput; a high-priority copy stream did not
    // change that). Two engines per process (2 + 2 streams) keep the overlap: 99.5 % of the
    // HBM-resident rate.
    if (!e->copy_stream) HIPCHK(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
    HIPCHK(hipHostMalloc((void**)&sl.h_res, sizeof(vt_result) * e->B));
    HIPCHK(hipHostMalloc((void**)&sl.h_st, sizeof(StreamState) * e->B));
    HIPCHK(hipEventCreateWithFlags(&sl.up_ev, hipEventDisableTiming));
    HIPCHK(hipEventCreateWithFlags(&sl.done_ev, hipEventDisableTiming));
    return VT_OK;
}

// exact (non-speculative) synchronous pass over `host` with the states the device holds now; results
// and states land in the slot's buffers
static int host_pass_exact_sync(Engine* e, Engine::HostSlot& sl) {
    const int n = e->B;
    std::vector<vt_frame> dev((size_t)n);
    std::vector<float> boxes((size_t)n * 4);
    for (int b = 0; b < n; ++b) memcpy(&boxes[(size_t)b * 4], e->known[b].box, 4 * sizeof(float));
    if (int rc = stage_host_frames(e, sl.host.data(), n, reinterpret_cast<const float(*)[4]>(boxes.data()), dev.data()))
        return rc;
    if (int rc = e->enqueue(dev.data(), n)) return rc;
    HIPCHK(hipStreamSynchronize(e->stream));
    memcpy(sl.h_res, e->h_results, sizeof(vt_result) * n);
    memcpy(sl.h_st, e->h_states_all, sizeof(StreamState) * n);
    for (int b = 0; b < n; ++b) e->known[b] = e->h_states_all[b];
    sl.redone = true;
    return VT_OK;
}

int vt_group_enqueue_host(vt_group* g, const vt_frame* host_frames, int n) try {
    if (!g || !host_frames) return set_err(VT_ERR_INVALID_ARG, "null argument");
    Engine* e = g->e;
    if (n != e->B) return set_err(VT_ERR_INVALID_ARG, "enqueue_host: need exactly %d frames", e->B);
    for (int b = 0; b < n; ++b)
        if (!e->h_initialized[b]) return set_err(VT_ERR_NOT_INITIALIZED, "stream %d not initialised", b);
    const unsigned outstanding = e->host_seq - e->host_collected;
    if (outstanding >= 2)
        return set_err(VT_ERR_INVALID_ARG, "enqueue_host: two passes outstanding, call vt_group_wait_next first");
    DEVICE_SCOPE(e->device);
    Engine::HostSlot& sl = e->hs[e->host_seq & 1];
    if (int rc = host_slot_prepare(e, sl)) return rc;
    if (outstanding == 0) {
        // nothing of ours is running: make sure nothing else is either, then the boxes are exact
        if (int rc = e->wait(nullptr, 0)) return rc;
    }
    sl.host.assign(host_frames, host_frames + n);
    sl.speculative = outstanding == 1;
    sl.redone = false;
    std::vector<vt_frame> dev((size_t)n);
    std::vector<float> boxes((size_t)n * 4);
    for (int b = 0; b < n; ++b) memcpy(&boxes[(size_t)b * 4], e->known[b].box, 4 * sizeof(float));
    const Arena a{&sl.d_arena, &sl.h_arena, &sl.bytes, e->copy_stream};
    if (int rc = stage_host_frames_to(e, a, host_frames, n, reinterpret_cast<const float(*)[4]>(boxes.data()),
                                      sl.speculative ? e->margin : 0.0f, dev.data(), nullptr))
        return rc;
    HIPCHK(hipEventRecord(sl.up_ev, e->copy_stream));
    HIPCHK(hipStreamWaitEvent(e->stream, sl.up_ev, 0));          // the pass starts behind ITS upload only
    if (int rc = e->enqueue(dev.data(), n, sl.h_res, sl.h_st)) return rc;   // results land in THIS slot's buffers
    HIPCHK(hipEventRecord(sl.done_ev, e->stream));
    sl.pending = true;
    e->host_seq += 1;
    return VT_OK;
} VT_NOTHROW_INT

int vt_group_wait_next(vt_group* g, vt_result* out, int n) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    Engine* e = g->e;
    if (e->host_seq == e->host_collected) return set_err(VT_ERR_INVALID_ARG, "wait_next: no pass outstanding");
    DEVICE_SCOPE(e->device);
    Engine::HostSlot& sl = e->hs[e->host_collected & 1];
    Engine::HostSlot& younger = e->hs[(e->host_collected + 1) & 1];
    const bool has_younger = e->host_seq - e->host_collected == 2;
    if (!sl.redone) {
        HIPCHK(hipEventSynchronize(sl.done_ev));
        bool miss = false;
        if (sl.speculative)
            for (int b = 0; b < e->B; ++b)
                miss = miss || (sl.h_st[b].window_miss != 0 && sl.h_st[b].window_miss == sl.h_st[b].frames_done);
        if (miss) {
            // a stream moved out of its speculative window: rewind to the states this pass started
            // from - `known`, the host's copy of the states the previous pass left (collected by the
            // wait_next before this one) - and redo it, and the pass queued behind it, which consumed
            // its wrong states, with exact windows
            e->host_redos += 1;
            HIPCHK(hipStreamSynchronize(e->stream));
            HIPCHK(hipMemcpy(e->d_states, e->known.data(), sizeof(StreamState) * e->B, hipMemcpyHostToDevice));
            if (int rc = host_pass_exact_sync(e, sl)) return rc;
            if (has_younger)
                if (int rc = host_pass_exact_sync(e, younger)) return rc;
        }
    }
    if (out)
        for (int b = 0; b < std::min(n, e->B); ++b) out[b] = sl.h_res[b];
    // boxes the next window is planned around: this pass's - unless a younger pass was redone just
    // now, whose states are newer (host_pass_exact_sync set `known` already)
    if (!(has_younger && younger.redone)) {
        for (int b = 0; b < e->B; ++b) e->known[b] = sl.h_st[b];
        memcpy(e->h_states_all, sl.h_st, sizeof(StreamState) * e->B);   // the engine's own mirrors follow
        memcpy(e->h_results, sl.h_res, sizeof(vt_result) * e->B);
    }
    sl.pending = false;
    e->host_collected += 1;
    return VT_OK;
} VT_NOTHROW_INT

int vt_group_host_redos(const vt_group* g) { return g ? (int)g->e->host_redos : 0; }

static int do_init(vt_tracker* t, const vt_frame* f, vt_bbox box) { return t->e->init_stream(0, f, box); }
static int do_update(vt_tracker* t, const vt_frame* f, vt_result* out) {
    if (!out) return set_err(VT_ERR_INVALID_ARG, "null result pointer");
    memset(out, 0, sizeof(*out));
    if (int rc = t->e->enqueue(f, 1)) return rc;
    return t->e->wait(out, 1);
}

int vt_init_rgb8(vt_tracker* t, const uint8_t* rgb, int w, int h, int stride_bytes, vt_bbox box) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f;
    const float fb[4] = {(float)box.x, (float)box.y, (float)box.width, (float)box.height};
    if (int rc = stage_host_frame(t->e, VT_PIX_RGB8, rgb, nullptr, w, h, stride_bytes, 0, fb, &f)) return rc;
    return do_init(t, &f, box);
} VT_NOTHROW_INT
int vt_update_rgb8(vt_tracker* t, const uint8_t* rgb, int w, int h, int stride_bytes, vt_result* out) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    if (!t->e->h_initialized[0]) return set_err(VT_ERR_NOT_INITIALIZED, "update before init");
    vt_frame f;
    if (int rc = stage_host_frame(t->e, VT_PIX_RGB8, rgb, nullptr, w, h, stride_bytes, 0, t->e->known[0].box, &f)) return rc;
    return do_update(t, &f, out);
} VT_NOTHROW_INT
int vt_init_yuy2(vt_tracker* t, const uint8_t* yuy2, int w, int h, int stride_bytes, vt_bbox box) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f;
    const float fb[4] = {(float)box.x, (float)box.y, (float)box.width, (float)box.height};
    if (int rc = stage_host_frame(t->e, VT_PIX_YUY2, yuy2, nullptr, w, h, stride_bytes, 0, fb, &f)) return rc;
    return do_init(t, &f, box);
} VT_NOTHROW_INT
int vt_update_yuy2(vt_tracker* t, const uint8_t* yuy2, int w, int h, int stride_bytes, vt_result* out) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    if (!t->e->h_initialized[0]) return set_err(VT_ERR_NOT_INITIALIZED, "update before init");
    vt_frame f;
    if (int rc = stage_host_frame(t->e, VT_PIX_YUY2, yuy2, nullptr, w, h, stride_bytes, 0, t->e->known[0].box, &f)) return rc;
    return do_update(t, &f, out);
} VT_NOTHROW_INT
int vt_init_nv12(vt_tracker* t, const uint8_t* y, const uint8_t* uv, int w, int h, int y_stride,
                 int uv_stride, vt_bbox box) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f;
    const float fb[4] = {(float)box.x, (float)box.y, (float)box.width, (float)box.height};
    if (int rc = stage_host_frame(t->e, VT_PIX_NV12, y, uv, w, h, y_stride, uv_stride, fb, &f)) return rc;
    return do_init(t, &f, box);
} VT_NOTHROW_INT
int vt_update_nv12(vt_tracker* t, const uint8_t* y, const uint8_t* uv, int w, int h, int y_stride,
                   int uv_stride, vt_result* out) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    if (!t->e->h_initialized[0]) return set_err(VT_ERR_NOT_INITIALIZED, "update before init");
    vt_frame f;
    if (int rc = stage_host_frame(t->e, VT_PIX_NV12, y, uv, w, h, y_stride, uv_stride, t->e->known[0].box, &f)) return rc;
    return do_update(t, &f, out);
} VT_NOTHROW_INT

static vt_frame dev_frame(int fmt, const void* p0, const void* p1, int w, int h, int s0, int s1) {
    vt_frame f;
    memset(&f, 0, sizeof(f));
    f.plane0 = p0; f.plane1 = p1; f.width = w; f.height = h; f.stride0 = s0; f.stride1 = s1;
    f.format = fmt;
    return f;
}
int vt_init_rgb8_device(vt_tracker* t, const void* d_rgb, int w, int h, int stride_bytes, vt_bbox box) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f = dev_frame(VT_PIX_RGB8, d_rgb, nullptr, w, h, stride_bytes, 0);
    return do_init(t, &f, box);
} VT_NOTHROW_INT
int vt_update_rgb8_device(vt_tracker* t, const void* d_rgb, int w, int h, int stride_bytes, vt_result* out) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f = dev_frame(VT_PIX_RGB8, d_rgb, nullptr, w, h, stride_bytes, 0);
    return do_update(t, &f, out);
} VT_NOTHROW_INT
int vt_init_nv12_device(vt_tracker* t, const void* d_y, const void* d_uv, int w, int h, int y_stride,
                        int uv_stride, vt_bbox box) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f = dev_frame(VT_PIX_NV12, d_y, d_uv, w, h, y_stride, uv_stride);
    return do_init(t, &f, box);
} VT_NOTHROW_INT
int vt_update_nv12_device(vt_tracker* t, const void* d_y, const void* d_uv, int w, int h, int y_stride,
                          int uv_stride, vt_result* out) try {
    if (!t) return set_err(VT_ERR_INVALID_ARG, "null tracker");
    vt_frame f = dev_frame(VT_PIX_NV12, d_y, d_uv, w, h, y_stride, uv_stride);
    return do_update(t, &f, out);
} VT_NOTHROW_INT

// a single tracker viewed as a group of one (taps, profiling, stream handle)
vt_group* vt_tracker_as_group(vt_tracker* t) try {
    return t ? &t->view : nullptr;   // owned by the tracker: two trackers never share a view
} VT_NOTHROW_PTR

// ---- reference colour converter ------------------------------------------------------------------------

static size_t nv12_bytes_read(size_t w, size_t h) {
    if (!w || !h) return 0;
    const size_t uv_rows = (h + 1) / 2;
    const size_t last = (uv_rows - 1) * w + ((w & 1) ? w : w - 1);
    return w * h + last + 1;
}

int vt_nv12_to_rgb8_device(int device_id, const void* d_nv12, size_t len, int w, int h, void* d_rgb_out,
                           void* hip_stream) try {
    if (!d_nv12 || !d_rgb_out || w <= 0 || h <= 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    hipStream_t st = (hipStream_t)hip_stream;
    if (len < (size_t)w * h * 3 / 2) {  // src/nv12_convert.rs:48-50: short buffer -> zero frame
        HIPCHK(hipMemsetAsync(d_rgb_out, 0, (size_t)w * h * 3, st));
        return VT_OK;
    }
    if (len < nv12_bytes_read(w, h))
        return set_err(VT_ERR_SHORT_BUFFER, "nv12 buffer of %zu bytes is shorter than the %zu the "
                       "conversion of a %dx%d frame reads", len, nv12_bytes_read(w, h), w, h);
    HIPCHK(launch_nv12_to_rgb8((const uint8_t*)d_nv12, w, h, (uint8_t*)d_rgb_out, st));
    return VT_OK;
} VT_NOTHROW_INT

int vt_nv12_to_rgb8(int device_id, const uint8_t* nv12, size_t len, int w, int h, uint8_t* rgb_out) try {
    if (!nv12 || !rgb_out || w <= 0 || h <= 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    const size_t out_bytes = (size_t)w * h * 3;
    if (len < (size_t)w * h * 3 / 2) {
        memset(rgb_out, 0, out_bytes);
        return VT_OK;
    }
    if (len < nv12_bytes_read(w, h))
        return set_err(VT_ERR_SHORT_BUFFER, "nv12 buffer too short for a %dx%d frame", w, h);
    uint8_t *din = nullptr, *dout = nullptr;
    HIPCHK(hipMalloc((void**)&din, len));
    hipError_t e = hipMalloc((void**)&dout, out_bytes);
    if (e != hipSuccess) { (void)hipFree(din); return set_err(VT_ERR_HIP, "hipMalloc: %s", hipGetErrorString(e)); }
    int rc = VT_OK;
    if ((e = hipMemcpy(din, nv12, len, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = launch_nv12_to_rgb8(din, w, h, dout, nullptr)) != hipSuccess ||
        (e = hipMemcpy(rgb_out, dout, out_bytes, hipMemcpyDeviceToHost)) != hipSuccess)
        rc = set_err(VT_ERR_HIP, "nv12_to_rgb8: %s", hipGetErrorString(e));
    (void)hipFree(din);
    (void)hipFree(dout);
    return rc;
} VT_NOTHROW_INT

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 4); }
};

// ---- overlays ---------------------------------------------------------------------------------------------

static int overlay_device(int device_id, void* d_surf, int width, int height, int stride, int min_stride,
                          const vt_draw_cmd* cmds, int n, void* hip_stream, bool rgb) {
    if (!d_surf || width <= 0 || height <= 0 || stride < min_stride || n < 0 || (n > 0 && !cmds))
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (n == 0) return VT_OK;
    if (n > 256) return set_err(VT_ERR_INVALID_ARG, "at most 256 draw commands per call");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    hipStream_t st = (hipStream_t)hip_stream;
    vt_draw_cmd* d_cmds = nullptr;
    HIPCHK(hipMallocAsync((void**)&d_cmds, sizeof(vt_draw_cmd) * n, st));
    // pageable source: hipMemcpyAsync has consumed `cmds` when it returns
    hipError_t e = hipMemcpyAsync(d_cmds, cmds, sizeof(vt_draw_cmd) * n, hipMemcpyHostToDevice, st);
    if (e == hipSuccess)
        e = rgb ? launch_overlay_rgb((uint8_t*)d_surf, width, height, stride, d_cmds, n, st)
                : launch_overlay((uint8_t*)d_surf, width, height, stride, d_cmds, n, st);
    (void)hipFreeAsync(d_cmds, st);
    if (e != hipSuccess) return set_err(VT_ERR_HIP, "overlay: %s", hipGetErrorString(e));
    return VT_OK;
}

static int overlay_host(int device_id, uint8_t* surf, size_t bytes, int width, int height, int stride,
                        const vt_draw_cmd* cmds, int n, bool rgb) {
    if (!surf || width <= 0 || height <= 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    DevBuf d;
    HIPCHK(d.alloc(bytes));
    HIPCHK(hipMemcpy(d.p, surf, bytes, hipMemcpyHostToDevice));
    if (int rc = overlay_device(device_id, d.p, width, height, stride, stride, cmds, n, nullptr, rgb)) return rc;
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(surf, d.p, bytes, hipMemcpyDeviceToHost));
    return VT_OK;
}

int vt_overlay_nv12_device(int device_id, void* d_y, int width, int height, int stride, const vt_draw_cmd* cmds,
                           int n, void* hip_stream) try {
    return overlay_device(device_id, d_y, width, height, stride, width, cmds, n, hip_stream, false);
} VT_NOTHROW_INT
int vt_overlay_nv12(int device_id, uint8_t* nv12, int width, int height, const vt_draw_cmd* cmds, int n) try {
    return overlay_host(device_id, nv12, (size_t)width * height, width, height, width, cmds, n, false);
} VT_NOTHROW_INT
int vt_overlay_rgb8_device(int device_id, void* d_rgb, int width, int height, int stride, const vt_draw_cmd* cmds,
                           int n, void* hip_stream) try {
    return overlay_device(device_id, d_rgb, width, height, stride, width * 3, cmds, n, hip_stream, true);
} VT_NOTHROW_INT
int vt_overlay_rgb8(int device_id, uint8_t* rgb, int width, int height, const vt_draw_cmd* cmds, int n) try {
    return overlay_host(device_id, rgb, (size_t)width * height * 3, width, height, width * 3, cmds, n, true);
} VT_NOTHROW_INT

// ---- operator-level entry points ---------------------------------------------------------------------------

// host-side helpers of the operator entry points: float32 <-> the bf16 pair of the residual stream
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
// row per output row) - the X-epilogues: c_inout goes in and comes back through the bf16 pair (hi + lo, 17
// significant bits), rowstat_out (if given) receives the finalized row terms (rstd, -mean * rstd) of x;
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
    g.M = M; g.N = N; g.K = K; g.Xh = (bf16_t*)dxh.p; g.Xl = (bf16_t*)dxl.p; g.ldx = N; g.Cb = (bf16_t*)dcb.p; g.ldcb = N;
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
    std::vector<bf16_t> hi, lo;
    if (epi == EPI_RESID) {
        hi.resize(MN); lo.resize(MN);
        for (size_t i = 0; i < MN; ++i) { hi[i] = host_bf16(c_inout[i]); lo[i] = host_bf16(c_inout[i] - host_f32(hi[i])); }
        HIPCHK(hipMemcpy(dxh.p, hi.data(), MN * 2, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(dxl.p, lo.data(), MN * 2, hipMemcpyHostToDevice));
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
    if (x_epi && rowstat_out) {
        HIPCHK(dro.alloc((size_t)M * 8));
        HIPCHK(dcnt.alloc((size_t)((M + 255) / 256 + 1) * 4));
        HIPCHK(hipMemset(dcnt.p, 0, (size_t)((M + 255) / 256 + 1) * 4));
        HIPCHK(hipMemset(dro.p, 0xff, (size_t)M * 8));
        const int eff = cfg < 0 ? gemm_effective_config(g, epi) : cfg;
        if (eff >= GEMM_CFG_256P8) {
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
        HIPCHK(hipMemcpy(lo.data(), dxl.p, MN * 2, hipMemcpyDeviceToHost));
        for (size_t i = 0; i < MN; ++i) c_inout[i] = host_f32(hi[i]) + host_f32(lo[i]);
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
        g.Xh = (bf16_t*)dcb.p; g.Xl = (bf16_t*)dxl.p; g.ldx = N; g.cstat = (float2*)dcst.p;
        HIPCHK(dcnt.alloc((size_t)((M + 255) / 256 + 1) * 4)); HIPCHK(dro.alloc((size_t)M * 8 + 16));
        HIPCHK(hipMemset(dcnt.p, 0, (size_t)((M + 255) / 256 + 1) * 4));
        g.rowstat_out = (float2*)dro.p; g.panel_cnt = (unsigned*)dcnt.p; g.ln_eps = 1e-6f;
    }
    else if (epilogue == EPI_QKV || epilogue == EPI_GELU_BF16) { g.rowstat = (const float2*)drs.p; g.colsum = (const float*)db.p; }
    g.pos = (const float*)dc.p; g.pos_rows = M;
    g.qk = (bf16_t*)dcb.p; g.vt = (bf16_t*)dvt.p; g.tokens = tokens; g.npad = npad; g.D = D;
    if (epilogue == EPI_QKV && (N % 192 || tokens != M)) return set_err(VT_ERR_INVALID_ARG, "qkv bench: N = 3D, D % 64 == 0, M % 4 == 0");
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
int vt_op_headconv_ln_bf16(int device_id, const uint16_t* xh, const uint16_t* xl, const float* gamma, const float* beta,
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
        HIPCHK(hipMemcpy(dl.p, xl, Mx * D * 2, hipMemcpyHostToDevice));
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
        h.xh = (const bf16_t*)dh.p; h.xl = (const bf16_t*)dl.p; h.ln_g = (const float*)dg.p; h.ln_b = (const float*)dbt.p;
        h.ln_eps = eps; h.in_stride = ntok; h.in_off = off;
    } else {
        h.in = (const bf16_t*)dfeat.p; h.ldin = D;
    }
    auto run = [&]() -> hipError_t {
        if (!fused) {
            hipError_t e = launch_layernorm_split((const bf16_t*)dh.p, (const bf16_t*)dl.p, (const float*)dg.p, (const float*)dbt.p,
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

// ---- RCCL start-up broadcast (librccl loaded lazily) -----------------------------------------------

namespace {
struct NcclId { char internal[VT_RCCL_ID_BYTES]; };   // ≙ ncclUniqueId
typedef void* NcclComm;
struct RcclApi {
    int (*GetUniqueId)(NcclId*) = nullptr;
    int (*CommInitRank)(NcclComm*, int, NcclId, int) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
    int (*CommDestroy)(NcclComm) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};
RcclApi load_rccl() {
    RcclApi api;
    void* h = RTLD_DEFAULT;                          // a copy already loaded by the host wins
    if (!dlsym(h, "ncclGetUniqueId")) {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        h = nullptr;
        for (const char* n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!h) return api;
    }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(h, "ncclCommInitRank");
    api.Broadcast = (decltype(api.Broadcast))dlsym(h, "ncclBroadcast");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(h, "ncclCommDestroy");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(h, "ncclGetErrorString");
    api.ok = api.GetUniqueId && api.CommInitRank && api.Broadcast && api.CommDestroy;
    return api;
}
// one host thread per GPU may call in at the same time: a C++11 magic static hands every caller the
// fully built table (initialisation runs once, the others wait for it)
RcclApi* rccl_api() {
    static RcclApi api = load_rccl();
    return api.ok ? &api : nullptr;
}
int rccl_err(RcclApi* a, const char* what, int code) {
    return set_err(VT_ERR_HIP, "%s failed: %s (ncclResult %d)", what,
                   a->GetErrorString ? a->GetErrorString(code) : "?", code);
}
}  // namespace

extern "C" {

int vt_rccl_unique_id(uint8_t id_out[VT_RCCL_ID_BYTES]) try {
    if (!id_out) return set_err(VT_ERR_INVALID_ARG, "null id buffer");
    RcclApi* a = rccl_api();
    if (!a) return set_err(VT_ERR_NO_DEVICE, "librccl could not be loaded (dlopen librccl.so.1 / librccl.so)");
    NcclId id;
    memset(&id, 0, sizeof(id));
    if (int rc = a->GetUniqueId(&id)) return rccl_err(a, "ncclGetUniqueId", rc);
    memcpy(id_out, &id, sizeof(id));
    return VT_OK;
} VT_NOTHROW_INT

int vt_broadcast_weights_rccl(const uint8_t id[VT_RCCL_ID_BYTES], int world, int rank, int device_id,
                              const char* weights_path, void** d_blob_out, size_t* bytes_out) try {
    if (!id || !d_blob_out || !bytes_out || world < 1 || rank < 0 || rank >= world)
        return set_err(VT_ERR_INVALID_ARG, "bad argument");
    *d_blob_out = nullptr; *bytes_out = 0;
    if (rank == 0 && !weights_path) return set_err(VT_ERR_INVALID_ARG, "rank 0 needs the weights path");
    if (int rc = check_device(device_id)) return rc;
    RcclApi* a = rccl_api();
    if (!a) return set_err(VT_ERR_NO_DEVICE, "librccl could not be loaded");
    DEVICE_SCOPE(device_id);
    std::vector<uint8_t> blob;
    if (rank == 0)
        if (int rc = read_file(weights_path, &blob)) return rc;
    NcclId nid;
    memcpy(&nid, id, sizeof(nid));
    NcclComm comm = nullptr;
    if (int rc = a->CommInitRank(&comm, world, nid, rank)) return rccl_err(a, "ncclCommInitRank", rc);
    hipStream_t st = nullptr;
    unsigned long long* d_n = nullptr;
    void* d_blob = nullptr;
    int ret = VT_OK;
    auto fail = [&](int code) { ret = code; };
    do {
        hipError_t he = hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
        if (he != hipSuccess) { fail(set_err(VT_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(he))); break; }
        if ((he = hipMalloc((void**)&d_n, 8)) != hipSuccess) { fail(set_err(VT_ERR_OOM, "hipMalloc: %s", hipGetErrorString(he))); break; }
        unsigned long long n = blob.size();
        if ((he = hipMemcpyAsync(d_n, &n, 8, hipMemcpyHostToDevice, st)) != hipSuccess) { fail(set_err(VT_ERR_HIP, "copy: %s", hipGetErrorString(he))); break; }
        if (int rc = a->Broadcast(d_n, d_n, 8, /*ncclUint8*/ 1, 0, comm, st)) { fail(rccl_err(a, "ncclBroadcast(size)", rc)); break; }
        if ((he = hipMemcpyAsync(&n, d_n, 8, hipMemcpyDeviceToHost, st)) != hipSuccess ||
            (he = hipStreamSynchronize(st)) != hipSuccess) { fail(set_err(VT_ERR_HIP, "size read-back: %s", hipGetErrorString(he))); break; }
        if (n < kHeaderBytes || n > (1ull << 36)) { fail(set_err(VT_ERR_FORMAT, "broadcast blob size %llu out of range", n)); break; }
        if ((he = hipMalloc(&d_blob, n)) != hipSuccess) { fail(set_err(VT_ERR_OOM, "hipMalloc(%llu): %s", n, hipGetErrorString(he))); break; }
        if (rank == 0 && (he = hipMemcpyAsync(d_blob, blob.data(), n, hipMemcpyHostToDevice, st)) != hipSuccess) {
            fail(set_err(VT_ERR_HIP, "blob upload: %s", hipGetErrorString(he))); break; }
        // one message: a single large transfer suits xGMI's per-link bandwidth better than many small ones
        if (int rc = a->Broadcast(d_blob, d_blob, n, 1, 0, comm, st)) { fail(rccl_err(a, "ncclBroadcast(blob)", rc)); break; }
        if ((he = hipStreamSynchronize(st)) != hipSuccess) { fail(set_err(VT_ERR_HIP, "broadcast: %s", hipGetErrorString(he))); break; }
        *d_blob_out = d_blob; *bytes_out = (size_t)n;
        d_blob = nullptr;
    } while (0);
    if (d_blob) (void)hipFree(d_blob);
    if (d_n) (void)hipFree(d_n);
    if (st) (void)hipStreamDestroy(st);
    (void)a->CommDestroy(comm);
    return ret;
} VT_NOTHROW_INT

void vt_free_device_blob(int device_id, void* d_blob) try {
    if (!d_blob) return;
    DeviceScope ds(device_id);
    (void)hipFree(d_blob);
} VT_NOTHROW_VOID

}  // extern "C"
