// vt_ingest.hip — host-frame ingest behind the C ABI: only the window of a frame that the pass can sample crosses
// PCIe; the pipelined form uploads frame t+1 on a copy stream while pass t runs.
#include "vt_engine.hpp"

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
int stage_host_frames(Engine* e, const vt_frame* host, int n, const float (*boxes)[4], vt_frame* dev) {
    const Arena a{&e->d_stage, &e->h_pack, &e->stage_bytes, e->stream};
    return stage_host_frames_to(e, a, host, n, boxes, 0.0f, dev, nullptr);
}

int stage_host_frame(Engine* e, int fmt, const uint8_t* p0, const uint8_t* p1, int w, int h,
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

extern "C" {

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


}  // extern "C"
