// vt_abi.hip — the extern "C" boundary of include/vittrack_hip.h (every entry a function-try-block: nothing unwinds
// into the host, /root/reference/Cargo.toml:37 panic = "abort").
#include "vt_engine.hpp"
#include <sys/stat.h>
#include <unistd.h>

extern "C" {

void vt_config_default(vt_config* cfg) try {
    if (!cfg) return;
    memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = sizeof(vt_config);
    cfg->success_threshold = -1.0f;
    cfg->use_graph = 1;
    cfg->n_streams = 1;
} VT_NOTHROW_VOID
const char* vt_last_error(void) { return vt_err_text(); }
int vt_abi_version(void) { return VT_ABI_VERSION; }
const char* vt_build_info(void) {
    static char buf[256];
    static std::once_flag once;
    std::call_once(once, [] { snprintf(buf, sizeof(buf), "abi=%d;k_gemm256=%s", VT_ABI_VERSION, gemm256_build_id()); });
    return buf;
}
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
    // the duplicate descriptor the import was given. ROCm 7.2 neither closes it at hipImportExternalMemory nor at
    // hipDestroyExternalMemory (tools/dmabuf_fd_probe.py: it is still open after both - every import leaked one descriptor
    // and with it a reference on the buffer), so the release closes it - but only while it still names the same open file
    // (device and inode as at import), in case another runtime version closes it itself and the number has been reused
    int fd;
    dev_t st_dev;
    ino_t st_ino;
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
        (void)close(dupfd);         // nothing opened a descriptor in between: at worst EBADF
        return set_err(VT_ERR_HIP, "hipExternalMemoryGetMappedBuffer: %s", hipGetErrorString(he));
    }
    struct stat sb;
    const bool have_id = fstat(dupfd, &sb) == 0;
    vt_extmem* xm = new (std::nothrow) vt_extmem{device_id, mem, have_id ? dupfd : -1, have_id ? sb.st_dev : 0, have_id ? sb.st_ino : 0};
    if (!xm) { (void)hipDestroyExternalMemory(mem); (void)close(dupfd); return set_err(VT_ERR_OOM, "out of host memory"); }
    *out = xm;
    *d_ptr = p;
    return VT_OK;
} VT_NOTHROW_INT

void vt_release_dmabuf(vt_extmem* m) try {
    if (!m) return;
    DeviceScope ds(m->device);             // the caller's current device is restored on return
    (void)hipDeviceSynchronize();          // no kernel of ours may still read the mapping
    (void)hipDestroyExternalMemory(m->mem);
    if (m->fd >= 0) {
        struct stat sb;
        if (fstat(m->fd, &sb) == 0 && sb.st_dev == m->st_dev && sb.st_ino == m->st_ino) (void)close(m->fd);
    }
    delete m;
} VT_NOTHROW_VOID

int vt_export_dmabuf(int device_id, const void* d_ptr, size_t bytes, int* fd_out) try {
    if (!d_ptr || !fd_out || bytes == 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    *fd_out = -1;
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    {   // the handle names the allocation: a range that starts inside one is mapped back from the allocation's base by the
        // importer (ROCm 7.2) - other bytes than the caller meant. Refused here instead of aliased there.
        hipDeviceptr_t base = nullptr;
        size_t span = 0;
        if (hipMemGetAddressRange(&base, &span, (hipDeviceptr_t)d_ptr) != hipSuccess || base != (hipDeviceptr_t)d_ptr || bytes > span)
            return set_err(VT_ERR_INVALID_ARG, "export_dmabuf: the range must begin at the start of a device allocation and lie inside it");
    }
    int fd = -1;
    hipError_t he = hipMemGetHandleForAddressRange(&fd, (hipDeviceptr_t)d_ptr, bytes,
                                                   hipMemRangeHandleTypeDmaBufFd, 0);
    if (he != hipSuccess || fd < 0)
        return set_err(VT_ERR_HIP, "hipMemGetHandleForAddressRange(dma-buf): %s", hipGetErrorString(he));
    *fd_out = fd;
    return VT_OK;
} VT_NOTHROW_INT

}  // extern "C"

// Host ranges mapped by vt_host_register: the host-pointer entry points look a frame's planes up here and, when
// both lie in a mapped range of the engine's device, hand the kernels the mapped pointers (no window packing, no
// staging copy). A handful of entries; a mutex, because registration and tracking run on different threads.
struct HostRange { const uint8_t* host; size_t bytes; uint8_t* dev; int device; };
static std::mutex g_ranges_mu;
static std::vector<HostRange> g_ranges;
static std::atomic<int> g_ranges_n{0};
// the whole extent [p, p + bytes) must lie inside one mapped range: a frame that only starts in one is staged
const uint8_t* mapped_device_ptr(int device, const uint8_t* p, size_t bytes) {
    if (!p || bytes == 0 || g_ranges_n.load(std::memory_order_acquire) == 0) return nullptr;
    std::lock_guard<std::mutex> lk(g_ranges_mu);
    for (const HostRange& r : g_ranges)
        if (r.device == device && p >= r.host && p < r.host + r.bytes && bytes <= (size_t)(r.host + r.bytes - p))
            return r.dev + (p - r.host);
    return nullptr;
}

extern "C" {

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
}  // extern "C"
// A pipelined host pass (vt_group_enqueue_host) that has not been collected owns the stream states:
// its redo path rewinds to the host's copy of them (`known`). Everything that would advance or
// overwrite the states behind such a pass is refused until vt_group_wait_next has collected it.
int refuse_while_pipelined(const Engine* e, const char* what) {
    if (e->host_seq != e->host_collected)
        return set_err(VT_ERR_INVALID_ARG, "%s: collect the pipelined host passes first (vt_group_wait_next)", what);
    return VT_OK;
}

extern "C" {

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

int vt_group_enable_taps(vt_group* g, int enable) try {
    if (!g) return set_err(VT_ERR_INVALID_ARG, "null group");
    Engine* e = g->e;
    DEVICE_SCOPE(e->device);
    HIPCHK(hipStreamSynchronize(e->stream));
    if (enable && !e->d_taps)      // per slot: the hi and the lo half of the residual stream
        HIPCHK(dalloc0(&e->d_taps, (size_t)(e->d.L + 1) * e->tap_slot_bytes(), e->stream));
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
    // the captured passes hold the old choice: drop them and capture again here, not inside the next pass
    e->drop_graphs();
    return e->capture_all_graphs();
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
// the residual stream in float32: the value the 3-byte pair stands for
// (the 3-byte pair of specification v3: hi bf16 + lo8 * 2^-12)
static int64_t copy_out_pair(const bf16_t* dhi, const uint8_t* dlo, int64_t count, float* out, int64_t cap) {
    if (!out) return count;
    if (cap < count) return set_err(VT_ERR_INVALID_ARG, "read_tensor: capacity %lld < %lld", (long long)cap, (long long)count);
    std::vector<bf16_t> hi((size_t)count);
    std::vector<int8_t> lo((size_t)count);
    if (hipMemcpy(hi.data(), dhi, 2 * count, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(lo.data(), dlo, count, hipMemcpyDeviceToHost) != hipSuccess)
        return set_err(VT_ERR_HIP, "read_tensor: copy failed");
    for (int64_t i = 0; i < count; ++i) {
        const uint32_t uh = ((uint32_t)hi[i]) << 16;
        float fh;
        memcpy(&fh, &uh, 4);
        out[i] = fh + (float)lo[i] * VT_LO_Q;
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
    if (n == "graph_replays") {        // passes replayed so far per crop tier (diagnostics: which captured pass ran)
        if (!out) return Engine::TIERS;
        if (capacity < Engine::TIERS) return set_err(VT_ERR_INVALID_ARG, "read_tensor: capacity too small");
        for (int t = 0; t < Engine::TIERS; ++t) out[t] = (float)e->graph_replays[t];
        return Engine::TIERS;
    }
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
        const uint8_t* base = e->d_taps + (size_t)slot * e->tap_slot_bytes();
        const bf16_t* hi = reinterpret_cast<const bf16_t*>(base) + b * d.ntok * d.D;
        const uint8_t* lo = base + sizeof(bf16_t) * M * d.D + b * d.ntok * d.D;
        return copy_out_pair(hi, lo, (int64_t)d.ntok * d.D, out, capacity);
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

int vt_group_host_redos(const vt_group* g) { return g ? (int)g->e->host_redos : 0; }
int vt_group_graph_captures(const vt_group* g) { return g ? g->e->graph_captures : 0; }

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

int vt_nv12_to_rgb8_batch_device(int device_id, const void* const* d_nv12, const size_t* lens, int n, int w, int h,
                                 void* const* d_rgb_out, void* hip_stream) try {
    if (!d_nv12 || !lens || !d_rgb_out || n < 1 || n > 65536 || w <= 0 || h <= 0) return set_err(VT_ERR_INVALID_ARG, "bad argument");
    if (int rc = check_device(device_id)) return rc;
    DEVICE_SCOPE(device_id);
    std::vector<const uint8_t*> in((size_t)n);
    std::vector<uint8_t*> out((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!d_nv12[i] || !d_rgb_out[i]) return set_err(VT_ERR_INVALID_ARG, "frame %d: null pointer", i);
        out[(size_t)i] = (uint8_t*)d_rgb_out[i];
        if (lens[i] < (size_t)w * h * 3 / 2) { in[(size_t)i] = nullptr; continue; }   // src/nv12_convert.rs:48-50: zero frame
        if (lens[i] < nv12_bytes_read(w, h))
            return set_err(VT_ERR_SHORT_BUFFER, "frame %d: nv12 buffer of %zu bytes is shorter than the %zu the conversion "
                           "of a %dx%d frame reads", i, lens[i], nv12_bytes_read(w, h), w, h);
        in[(size_t)i] = (const uint8_t*)d_nv12[i];
    }
    HIPCHK(launch_nv12_to_rgb8_batch(in.data(), out.data(), n, w, h, (hipStream_t)hip_stream));
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
}  // extern "C"
