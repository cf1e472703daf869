// host_capi.cpp — extern "C" surface of the C++ host mirror (include/vittrack_host.h).
#include <cstring>
#include <string>

#include "../include/vittrack_host.h"
#include "tracker_context.hpp"

using host::TrackerContext;
using host::UserCommand;
using vit_tracker::BBox;
using vit_tracker::FrameView;

static thread_local std::string g_err;

struct vth_ctx { std::unique_ptr<TrackerContext> ctx; };
struct vth_timing { host::TimingStats stats; };

namespace {
struct CallbackTracker : vit_tracker::ITracker {
    vth_tracker_callbacks cb;
    void* user;
    CallbackTracker(vth_tracker_callbacks c, void* u) : cb(c), user(u) {}
    void init(const FrameView& f, BBox b) override {
        if (cb.init) cb.init(user, f.data, f.width, f.height, f.stride, (int)f.format, vt_bbox{b.x, b.y, b.width, b.height});
    }
    bool update(const FrameView& f, vit_tracker::TrackResult* out, std::string* err) override {
        vt_result r;
        memset(&r, 0, sizeof(r));
        if (!cb.update || cb.update(user, f.data, f.width, f.height, f.stride, (int)f.format, &r) != 0) {
            if (err) *err = "tracker callback failed";
            return false;
        }
        out->success = r.success != 0;
        out->score = r.score;
        out->bbox[0] = r.bbox.x; out->bbox[1] = r.bbox.y; out->bbox[2] = r.bbox.width; out->bbox[3] = r.bbox.height;
        return true;
    }
};
int give(const std::optional<BBox>& b, vt_bbox* out) {
    if (!b) return 0;
    if (out) *out = vt_bbox{b->x, b->y, b->width, b->height};
    return 1;
}
}  // namespace

extern "C" {

const char* vth_last_error(void) { return g_err.c_str(); }

int vth_ctx_new(const char* model_path, int width, int height, int device, vth_ctx** out) {
    if (!model_path || !out) { g_err = "null argument"; return VT_ERR_INVALID_ARG; }
    std::string err;
    auto ctx = TrackerContext::new_(model_path, width, height, device, &err);
    if (!ctx) { g_err = err; return VT_ERR_NO_DEVICE; }
    *out = new vth_ctx{std::move(ctx)};
    return VT_OK;
}
int vth_ctx_new_with_tracker(vth_tracker_callbacks cb, void* user, int width, int height, vth_ctx** out) {
    if (!out || !cb.update) { g_err = "null argument"; return VT_ERR_INVALID_ARG; }
    *out = new vth_ctx{TrackerContext::with_tracker(std::make_unique<CallbackTracker>(cb, user), width, height)};
    return VT_OK;
}
void vth_ctx_free(vth_ctx* c) { delete c; }
void vth_ctx_handle_command(vth_ctx* c, int command, int fast) {
    if (!c || command < 0 || command > 6) return;
    c->ctx->handle_command(UserCommand{(UserCommand::Kind)command, fast != 0});
}
int vth_ctx_process_frame_rgb8(vth_ctx* c, const uint8_t* rgb, int w, int h, int stride, vt_bbox* out) {
    if (!c || !rgb) { g_err = "null argument"; return VT_ERR_INVALID_ARG; }
    return give(c->ctx->process_frame(FrameView::rgb8(rgb, w, h, stride)), out);
}
int vth_ctx_process_frame_nv12(vth_ctx* c, const uint8_t* nv12, int w, int h, vt_bbox* out) {
    if (!c || !nv12) { g_err = "null argument"; return VT_ERR_INVALID_ARG; }
    return give(c->ctx->process_frame(FrameView::nv12_packed(nv12, w, h)), out);
}
const char* vth_ctx_state_name(const vth_ctx* c) { return c ? c->ctx->state_name() : ""; }
void vth_ctx_get_info(const vth_ctx* c, vth_ctx_info* o) {
    if (!c || !o) return;
    const TrackerContext& t = *c->ctx;
    memset(o, 0, sizeof(*o));
    o->state_kind = (int32_t)t.state.kind;
    o->lost_frames = t.state.frames;
    o->has_bbox = t.current_bbox ? 1 : 0;
    if (t.current_bbox) o->current_bbox = vt_bbox{t.current_bbox->x, t.current_bbox->y, t.current_bbox->width, t.current_bbox->height};
    o->current_score = t.current_score;
    o->pending_confirm = t.pending_confirm ? 1 : 0;
    o->cursor_x = t.selection.cursor_x; o->cursor_y = t.selection.cursor_y;
    o->start_x = t.selection.start_x; o->start_y = t.selection.start_y;
    o->selection_phase = (int32_t)t.selection.phase;
    o->frame_width = t.frame_width; o->frame_height = t.frame_height;
}

vt_bbox vth_selection_bbox(int start_x, int start_y, int cursor_x, int cursor_y) {
    host::SelectionState s = host::SelectionState::new_(0, 0);
    s.start_x = start_x; s.start_y = start_y; s.cursor_x = cursor_x; s.cursor_y = cursor_y;
    const BBox b = s.get_bbox();
    return vt_bbox{b.x, b.y, b.width, b.height};
}

vth_timing* vth_timing_new(void) { return new vth_timing(); }
void vth_timing_free(vth_timing* t) { delete t; }
void vth_timing_add_interval(vth_timing* t, uint64_t us) { if (t) t->stats.add_interval(us); }
void vth_timing_add_times(vth_timing* t, uint64_t conv_us, uint64_t track_us) { if (t) t->stats.add_times(conv_us, track_us); }
double vth_timing_fps(const vth_timing* t) { return t ? t->stats.fps() : 0.0; }
double vth_timing_avg_conv_ms(const vth_timing* t) { return t ? t->stats.avg_conv_ms() : 0.0; }
double vth_timing_avg_track_ms(const vth_timing* t) { return t ? t->stats.avg_track_ms() : 0.0; }

int vth_nv12_full_to_rgb(int device, const uint8_t* nv12, size_t len, int w, int h, uint8_t* rgb) {
    std::string err;
    vit_tracker::HipApi* api = vit_tracker::HipApi::get(&err);
    if (!api) { g_err = err; return VT_ERR_NO_DEVICE; }
    int rc = api->nv12_to_rgb8(device, nv12, len, w, h, rgb);
    if (rc != VT_OK) g_err = api->last_error();
    return rc;
}

}  // extern "C"
