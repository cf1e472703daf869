// vit_tracker.hpp — C++ mirror of the `vit_tracker` crate surface the reference host links
// (/root/reference/Cargo.toml:24): BBox, the update result, and VitTrack{new, init, update}, here
// backed by libvittrack_hip.so through its C ABI (include/vittrack_hip.h). The library is opened
// with dlopen when the first tracker is created, so this host code loads on machines without a
// GPU; creating a tracker there fails with the library's own error (no CPU fallback).
#pragma once
#include <dlfcn.h>

#include <cstdint>
#include <cstring>
#include <memory>
#include <string>

#include "../include/vittrack_hip.h"

namespace vit_tracker {

// ≙ vit_tracker::BBox — src/selection_state.rs:44 (BBox::new), src/tracker_context.rs:94
// (BBox::from_array), fields x/y/width/height read at src/tracker_context.rs:85, src/pipeline.rs:166
struct BBox {
    int32_t x = 0, y = 0, width = 0, height = 0;
    static BBox new_(int32_t x, int32_t y, int32_t w, int32_t h) { return BBox{x, y, w, h}; }
    static BBox from_array(const int32_t (&a)[4]) { return BBox{a[0], a[1], a[2], a[3]}; }
};

// ≙ the Ok value of VitTrack::update (fields used at src/tracker_context.rs:92-95,122-125)
struct TrackResult {
    bool success = false;
    float score = 0.0f;
    int32_t bbox[4] = {0, 0, 0, 0};
};

// ≙ &ArrayView3<u8> (H,W,3) RGB8 (src/pipeline_ir.rs:142) plus the fused NV12 form
struct FrameView {
    enum Format { RGB8 = 0, NV12 = 1 };
    Format format = RGB8;
    const uint8_t* data = nullptr;  // RGB8 pixels or Y plane
    const uint8_t* uv = nullptr;    // NV12 UV plane
    int width = 0, height = 0, stride = 0, uv_stride = 0;
    static FrameView rgb8(const uint8_t* p, int w, int h, int stride) {
        FrameView f; f.format = RGB8; f.data = p; f.width = w; f.height = h; f.stride = stride; return f;
    }
    // packed NV12 as the reference maps it: Y plane then UV, stride == width (src/nv12_convert.rs:47-54)
    static FrameView nv12_packed(const uint8_t* p, int w, int h) {
        FrameView f; f.format = NV12; f.data = p; f.uv = p + (size_t)w * h; f.width = w; f.height = h;
        f.stride = w; f.uv_stride = (w + 1) & ~1; return f;
    }
};

// What TrackerContext needs from a tracker (src/tracker_context.rs:88,90,120). `update` returns
// false for the reference's Err(_) arm and fills `err`.
struct ITracker {
    virtual ~ITracker() = default;
    virtual void init(const FrameView& full_image, BBox bbox) = 0;
    virtual bool update(const FrameView& full_image, TrackResult* out, std::string* err) = 0;
};

// C-ABI entry points resolved from libvittrack_hip.so
struct HipApi {
    void* handle = nullptr;
    decltype(&vt_create) create = nullptr;
    decltype(&vt_destroy) destroy = nullptr;
    decltype(&vt_last_error) last_error = nullptr;
    decltype(&vt_config_default) config_default = nullptr;
    decltype(&vt_init_rgb8) init_rgb8 = nullptr;
    decltype(&vt_update_rgb8) update_rgb8 = nullptr;
    decltype(&vt_init_nv12) init_nv12 = nullptr;
    decltype(&vt_update_nv12) update_nv12 = nullptr;
    decltype(&vt_nv12_to_rgb8) nv12_to_rgb8 = nullptr;

    static HipApi* get(std::string* err) {
        static HipApi api;
        static bool tried = false;
        static std::string load_err;
        if (!tried) {
            tried = true;
            Dl_info info;
            std::string dir;
            if (dladdr((void*)&HipApi::get, &info) && info.dli_fname) {
                dir = info.dli_fname;
                size_t p = dir.rfind('/');
                dir = p == std::string::npos ? "." : dir.substr(0, p);
            }
            const char* env = getenv("VITTRACK_HIP_LIB");
            std::string path = env ? env : dir + "/../gstreamer-vit-tracker_amd/libvittrack_hip.so";
            api.handle = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (!api.handle) {
                load_err = std::string("cannot load ") + path + ": " + dlerror();
            } else {
#define VT_SYM(field, name) api.field = (decltype(api.field))dlsym(api.handle, #name)
                VT_SYM(create, vt_create); VT_SYM(destroy, vt_destroy); VT_SYM(last_error, vt_last_error);
                VT_SYM(config_default, vt_config_default); VT_SYM(init_rgb8, vt_init_rgb8);
                VT_SYM(update_rgb8, vt_update_rgb8); VT_SYM(init_nv12, vt_init_nv12);
                VT_SYM(update_nv12, vt_update_nv12); VT_SYM(nv12_to_rgb8, vt_nv12_to_rgb8);
#undef VT_SYM
                if (!api.create || !api.destroy || !api.last_error || !api.init_rgb8 || !api.update_rgb8 ||
                    !api.init_nv12 || !api.update_nv12 || !api.nv12_to_rgb8 || !api.config_default)
                    load_err = "libvittrack_hip.so lacks expected symbols";
            }
        }
        if (!load_err.empty()) { if (err) *err = load_err; return nullptr; }
        return &api;
    }
};

// ≙ vit_tracker::VitTrack
class VitTrack : public ITracker {
  public:
    // ≙ VitTrack::new(model_path) -> Result<VitTrack, E> (src/tracker_context.rs:21)
    static std::unique_ptr<VitTrack> new_(const std::string& model_path, int device, std::string* err) {
        HipApi* api = HipApi::get(err);
        if (!api) return nullptr;
        vt_tracker* h = nullptr;
        vt_config cfg;
        api->config_default(&cfg);
        if (api->create(model_path.c_str(), device, &cfg, &h) != VT_OK) {
            if (err) *err = api->last_error();
            return nullptr;
        }
        return std::unique_ptr<VitTrack>(new VitTrack(api, h));
    }
    ~VitTrack() override { if (h_) api_->destroy(h_); }

    // ≙ tracker.init(full_image, bbox); the reference discards the result (src/tracker_context.rs:88)
    void init(const FrameView& f, BBox b) override {
        vt_bbox cb{b.x, b.y, b.width, b.height};
        int rc = f.format == FrameView::NV12
                     ? api_->init_nv12(h_, f.data, f.uv, f.width, f.height, f.stride, f.uv_stride, cb)
                     : api_->init_rgb8(h_, f.data, f.width, f.height, f.stride, cb);
        last_init_rc_ = rc;
    }
    // ≙ tracker.update(full_image) -> Result<{success, score, bbox}, E>
    bool update(const FrameView& f, TrackResult* out, std::string* err) override {
        vt_result r;
        int rc = f.format == FrameView::NV12
                     ? api_->update_nv12(h_, f.data, f.uv, f.width, f.height, f.stride, f.uv_stride, &r)
                     : api_->update_rgb8(h_, f.data, f.width, f.height, f.stride, &r);
        if (rc != VT_OK) {
            if (err) *err = api_->last_error();
            return false;
        }
        out->success = r.success != 0;
        out->score = r.score;
        out->bbox[0] = r.bbox.x; out->bbox[1] = r.bbox.y; out->bbox[2] = r.bbox.width; out->bbox[3] = r.bbox.height;
        return true;
    }
    int last_init_status() const { return last_init_rc_; }

  private:
    VitTrack(HipApi* api, vt_tracker* h) : api_(api), h_(h) {}
    HipApi* api_;
    vt_tracker* h_;
    int last_init_rc_ = 0;
};

}  // namespace vit_tracker
