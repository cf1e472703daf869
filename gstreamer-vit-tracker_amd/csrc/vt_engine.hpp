// vt_engine.hpp — what the host-side translation units of libvittrack_hip.so share: error plumbing, the device
// scope, the Engine (one GPU's batch of tracked streams) and the handles of the C ABI.
//   vt_engine.hip   the Engine: weight blob, buffers, the per-frame launch plan, graphs, enqueue / wait
//   vt_abi.hip      the extern "C" boundary of include/vittrack_hip.h (create / init / update, groups, diagnostics,
//                   colour converter, overlays, dma-buf and host-mapping ingest)
//   vt_ingest.hip   host-frame ingest: window planning and packing, the pipelined enqueue_host / wait_next
//   vt_rccl.hip     the start-up weight broadcast over a lazily loaded librccl
//   vt_ops.hip      operator-level entry points of include/vittrack_hip_ops.h - linked into
//                   libvittrack_hip_ops.so (tests, tuning tools) only, NOT into the product library
#pragma once
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

// ---- error plumbing ------------------------------------------------------------------------------

int set_err(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
char* vt_err_text();        // the calling thread's error text (512 bytes)
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
    // residual stream as the 3-byte pair (x = xh + xl * 2^-12), its chunk partial statistics and the row terms of the
    // folded LayerNorm (vt_common.hpp); folded weights of all layers
    bf16_t *d_xh = nullptr, *d_foldw = nullptr;
    uint8_t* d_xl = nullptr;                      // the low half of the pair: one signed byte per element (spec v3, vt_common.hpp)
    uint8_t* d_taps = nullptr;                    // [slot][hi: M*D bf16 | lo8: M*D bytes]
    size_t tap_slot_bytes() const { return (size_t)B * d.ntok * d.D * 3; }
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
    // one captured pass per crop-buffer tier (k_preproc.hip: 16 / 32 / 64 KiB of LDS per tile), all captured at creation
    static constexpr int TIERS = 3;
    hipGraph_t graph[TIERS] = {nullptr, nullptr, nullptr};
    hipGraphExec_t graph_exec[TIERS] = {nullptr, nullptr, nullptr};
    int crop_tier = 0;                            // tier of the pass being enqueued (from the boxes the host knows)
    int crop_tier_forced = -1;                    // >= 0: tests / A-B runs (vt_group_set_tuning "crop_tier")
    int graph_captures = 0;                       // hipGraph captures since creation (vt_group_graph_captures)
    long graph_replays[TIERS] = {0, 0, 0};        // passes replayed per tier (vt_group_read_tensor "graph_replays")
    bool head_band_ok = false;                    // the head's band kernel takes this model's shapes (planner consulted at creation)
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
    int capture_all_graphs();
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

int read_file(const char* path, std::vector<uint8_t>* out);
int check_device(int device_id);
int make_engine(const char* path, const void* d_src, size_t bytes, int device_id, const vt_config* cfg, int B, Engine** out);
void fill_info(const Engine* e, vt_model_info* o);
int check_frame(const vt_frame& f);
void to_desc(const vt_frame& f, FrameDesc* o);

// ---- handles of the C ABI --------------------------------------------------------------------------
struct vt_group { Engine* e; };
struct vt_tracker { Engine* e; vt_group view; };   // view: the tracker as a group of one

// A pipelined host pass (vt_group_enqueue_host) that has not been collected owns the stream states
int refuse_while_pipelined(const Engine* e, const char* what);

// ---- host-frame ingest (vt_ingest.hip) ---------------------------------------------------------------
int stage_host_frames(Engine* e, const vt_frame* host, int n, const float (*boxes)[4], vt_frame* dev);
int stage_host_frame(Engine* e, int fmt, const uint8_t* p0, const uint8_t* p1, int w, int h, int s0, int s1,
                     const float* box, vt_frame* f);
const uint8_t* mapped_device_ptr(int device, const uint8_t* p, size_t bytes);

struct DevBuf {
    void* p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t n) { return hipMalloc(&p, n ? n : 4); }
};
size_t nv12_bytes_read(size_t w, size_t h);
