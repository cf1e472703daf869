/*
 * vittrack_hip.h — C ABI of libvittrack_hip.so, the MI355X (gfx950) tracker hot path.
 *
 * This is the drop-in boundary for the reference's `vit_tracker` crate API
 * (`VitTrack::{new, init, update}`, `BBox`), which the reference host calls at
 *   - src/tracker_context.rs:21   VitTrack::new(model_path)        -> vt_create
 *   - src/tracker_context.rs:88   tracker.init(full_image, bbox)   -> vt_init_rgb8 / vt_init_nv12
 *   - src/tracker_context.rs:90   tracker.update(full_image)       -> vt_update_rgb8 / vt_update_nv12
 *   - src/tracker_context.rs:120  tracker.update(full_image)       -> vt_update_rgb8 / vt_update_nv12
 *   - src/selection_state.rs:44   BBox::new(x, y, w, h)            -> vt_bbox
 *   - src/tracker_context.rs:94   BBox::from_array(&result.bbox)   -> vt_result.bbox
 * and for the reference's own colour converter
 *   - src/nv12_convert.rs:46      nv12_full_to_rgb_parallel        -> vt_nv12_to_rgb8
 *
 * Rules that follow from the reference call sites:
 *   - plain pointers and sizes only; no C++/torch types cross this line;
 *   - every call is synchronous with respect to the caller's frame buffer: on
 *     return the library no longer reads it (the host draws overlays into the
 *     same buffer right after, src/pipeline.rs:125);
 *   - a handle has no thread affinity (created on the main thread, used on the
 *     GStreamer streaming thread, src/pipeline.rs:55-67); calls on ONE handle
 *     must be serialised by the caller (the reference holds a Mutex);
 *   - nothing throws or aborts across the boundary (release profile is
 *     panic="abort", Cargo.toml:37): every entry returns a vt_status code and
 *     vt_last_error() gives the text;
 *   - the accept gate `success && score > 0.25` stays on the caller side
 *     (src/tracker_context.rs:93,122).
 *
 * There is no CPU fallback behind this ABI: if no gfx950 device is present
 * vt_create fails with VT_ERR_NO_DEVICE.
 */
#ifndef VITTRACK_HIP_H
#define VITTRACK_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VT_ABI_VERSION 5   /* 5: vt_nv12_to_rgb8_batch_device, vt_group_graph_captures; the operator-level test hooks (vt_op_*) moved to vittrack_hip_ops.h / libvittrack_hip_ops.so - the product library exports this header's symbols only; 4: vt_config.host_zero_copy (a former reserved slot: zero = the old default for single trackers), vt_group_set_tuning, vt_op_headconv_bf16, vt_op_headconv_ln_bf16 - additions only, a host built against 3 keeps working; 3: vt_op_gemm_bf16 / vt_op_qkv_bf16 take folded-LayerNorm terms; 2: vt_frame.window_w/h, vt_config.max_device_mib, explicit cfg/mode on vt_op_* */

typedef enum vt_status {
    VT_OK = 0,
    VT_ERR_INVALID_ARG = -1,
    VT_ERR_NO_DEVICE = -2,
    VT_ERR_IO = -3,          /* weights file missing / unreadable */
    VT_ERR_FORMAT = -4,      /* weights blob malformed or unsupported shape */
    VT_ERR_HIP = -5,         /* HIP runtime error (text in vt_last_error) */
    VT_ERR_NOT_INITIALIZED = -6, /* update before init */
    VT_ERR_SHORT_BUFFER = -7,
    VT_ERR_OOM = -8
} vt_status;

/* ≙ vit_tracker::BBox { x, y, width, height: i32 } (src/selection_state.rs:44,
 * src/tracker_context.rs:85) */
typedef struct vt_bbox {
    int32_t x, y, width, height;
} vt_bbox;

/* ≙ the value returned by VitTrack::update: fields success / score / bbox as
 * used at src/tracker_context.rs:92-95,122-125 */
typedef struct vt_result {
    int32_t success;
    float score;
    vt_bbox bbox;
} vt_result;

typedef struct vt_config {
    uint32_t struct_size;      /* = sizeof(vt_config); lets the struct grow */
    float success_threshold;   /* result.success = score >= this; <0 → blob default (0.20) */
    int32_t use_graph;         /* 1 (default): replay the frame as a hipGraph; 0: eager launches */
    int32_t n_streams;         /* vt_group_create only: independent tracked streams on this GPU,
                                * 1..VT_MAX_STREAMS */
    int32_t max_frame_width;   /* staging size for host-pointer calls; 0 → 3840 */
    int32_t max_frame_height;  /* 0 → 2160 */
    int32_t max_device_mib;    /* > 0: refuse (VT_ERR_OOM) to create an engine whose weights +
                                * activations need more HBM than this; 0: only the device's free
                                * memory limits it (checked before anything is allocated) */
    int32_t host_window_margin_pct; /* vt_group_enqueue_host: enlargement of the speculative window in
                                * percent of the crop side; 0 -> 75 (see there); < 0 -> none (tests:
                                * every moving target then takes the redo path) */
    int32_t host_zero_copy;    /* host-pointer calls on frames inside a vt_host_register range: 0 (default) =
                                * zero-copy route on single-stream engines (vt_create, n_streams 1) only,
                                * batched engines keep packing windows; 1 = zero-copy on every engine;
                                * -1 = never (always pack + copy). Measured trade-off at vt_host_register */
    int32_t reserved[5];
} vt_config;
#define VT_MAX_STREAMS 1024

typedef struct vt_model_info {
    int32_t patch, template_size, search_size, dim, heads, layers, mlp_dim;
    int32_t head_channels, tokens_template, tokens_search, kpad;
    int32_t score_grid;        /* search_size / patch */
    double flops_per_frame;    /* algorithmic FLOPs of one update (encoder+patch+head) */
    double encoder_flops_per_frame; /* encoder + patch-embed only */
    uint64_t weight_bytes;
} vt_model_info;

typedef struct vt_tracker vt_tracker; /* one tracked stream  ≙ VitTrack */
typedef struct vt_group vt_group;     /* B independent streams batched on one GPU */

void vt_config_default(vt_config* cfg);
const char* vt_last_error(void);       /* thread-local text of the last failure */
int vt_abi_version(void);
/* "abi=5;k_gemm256=<sha256>;..." - identity of this build: the ABI version and, per kernel translation unit that a
 * committed measurement refers to, the sha256 over its sources and compile flags that build.py stamped into it
 * (bench.py ties profiles/r06_dominant_kernel_pmc.json to the kernel that is running through it). Static string. */
const char* vt_build_info(void);
int vt_device_count(void);             /* gfx950 devices visible; 0 → vt_create fails */
/* Streams per vt_group that fill the MI355X's 256 CUs in whole rounds of the 256x256 GEMM kernel
 * (the smallest batch <= max_streams whose worst encoder GEMM wastes < 2 % of its rounds; 1 if
 * there is none or the model's width does not fit that kernel). No reference counterpart: the
 * reference runs one tracker per process (src/pipeline.rs:55). Needs no GPU. */
int vt_recommended_streams(const vt_model_info* info, int max_streams);
/* How to spread n_streams over the vt_groups ("engines") of ONE GPU so that no engine pays for an
 * almost empty GEMM round: up to R = vt_recommended_streams() one engine; between R and 2R an engine
 * of R plus one with the rest (their kernels overlap on the chip: 31 streams cost 172 us per frame
 * as 30 + 1 against 197 in one engine and 164 at 30); from 2R on two engines of n/2 (three or more
 * concurrent engines measured worse than two). Writes the engine sizes (each <= VT_MAX_STREAMS) to
 * sizes[0..cap) and returns how many, 0 on bad arguments. No reference counterpart (one tracker per
 * process there, src/pipeline.rs:55). Needs no GPU. */
int vt_plan_engines(const vt_model_info* info, int n_streams, int* sizes, int cap);

/* ---- single stream: the literal drop-in ------------------------------------------------ */

/* ≙ VitTrack::new(model_path) (src/tracker_context.rs:21). */
int vt_create(const char* weights_path, int device_id, const vt_config* cfg, vt_tracker** out);
/* Same, with the weight blob already in this device's HBM (after the RCCL start-up broadcast,
 * SURVEY.md §8e). The library makes its own copy; the caller may free d_blob on return. */
int vt_create_from_device_blob(const void* d_blob, size_t bytes, int device_id,
                               const vt_config* cfg, vt_tracker** out);
void vt_destroy(vt_tracker* t);
int vt_get_model_info(const vt_tracker* t, vt_model_info* out);

/* ≙ tracker.init(&ArrayView3<u8>, bbox) with the (H,W,3) RGB8 view of src/pipeline_ir.rs:142 /
 * src/nv12_convert.rs:90: C-contiguous rows of `stride_bytes` (>= 3*w), channel order R,G,B. */
int vt_init_rgb8(vt_tracker* t, const uint8_t* rgb, int w, int h, int stride_bytes, vt_bbox box);
/* ≙ tracker.update(&ArrayView3<u8>) -> Result<{success, score, bbox}> */
int vt_update_rgb8(vt_tracker* t, const uint8_t* rgb, int w, int h, int stride_bytes,
                   vt_result* out);

/* Fused ingest: the host skips nv12_full_to_rgb_parallel (src/pipeline.rs:105); every pixel the
 * tracker samples goes through exactly the reference's integer conversion
 * (src/nv12_convert.rs:109-147). y/uv are the two NV12 planes; strides in bytes. */
int vt_init_nv12(vt_tracker* t, const uint8_t* y, const uint8_t* uv, int w, int h, int y_stride,
                 int uv_stride, vt_bbox box);
int vt_update_nv12(vt_tracker* t, const uint8_t* y, const uint8_t* uv, int w, int h,
                   int y_stride, int uv_stride, vt_result* out);

/* Fused YUY2 ingest (packed 4:2:2, stride in bytes >= 2*w, w even): the capture format of the
 * reference's live pipeline (src/pipeline_ir.rs:27-41), so the host can skip videoconvert. */
int vt_init_yuy2(vt_tracker* t, const uint8_t* yuy2, int w, int h, int stride_bytes, vt_bbox box);
int vt_update_yuy2(vt_tracker* t, const uint8_t* yuy2, int w, int h, int stride_bytes,
                   vt_result* out);

/* Same four calls with the frame already resident in this GPU's HBM (device pointers). */
int vt_init_rgb8_device(vt_tracker* t, const void* d_rgb, int w, int h, int stride_bytes,
                        vt_bbox box);
int vt_update_rgb8_device(vt_tracker* t, const void* d_rgb, int w, int h, int stride_bytes,
                          vt_result* out);
int vt_init_nv12_device(vt_tracker* t, const void* d_y, const void* d_uv, int w, int h,
                        int y_stride, int uv_stride, vt_bbox box);
int vt_update_nv12_device(vt_tracker* t, const void* d_y, const void* d_uv, int w, int h,
                          int y_stride, int uv_stride, vt_result* out);

/* ---- start-up weight broadcast over RCCL, for hosts without Python ------------------------------
 * The path shards by stream (one process / host thread per GPU, no per-frame exchange); its only
 * collective is the weight broadcast at start-up. librccl is loaded lazily
 * (dlopen; symbols already in the process - e.g. PyTorch's bundled copy - are used first), so the
 * library has no link-time dependency on it. Protocol, one caller per GPU:
 *   rank 0        vt_rccl_unique_id(id)           128 opaque bytes (ncclUniqueId); the HOST ships
 *                                                 them to the other ranks (file, socket, env, ...)
 *   every rank    vt_broadcast_weights_rccl(id, world, rank, device, path, &d_blob, &bytes)
 *                                                 rank 0 reads `path` (ignored elsewhere); one
 *                                                 ncclBroadcast of the size, one of the bytes
 *   every rank    vt_create_from_device_blob / vt_group_create_from_device_blob(d_blob, bytes, ...)
 *   every rank    vt_free_device_blob(device, d_blob)
 * VT_ERR_NO_DEVICE if librccl cannot be loaded, VT_ERR_HIP for RCCL errors (text in vt_last_error). */
#define VT_RCCL_ID_BYTES 128
int vt_rccl_unique_id(uint8_t id_out[VT_RCCL_ID_BYTES]);
int vt_broadcast_weights_rccl(const uint8_t id[VT_RCCL_ID_BYTES], int world, int rank, int device_id,
                              const char* weights_path, void** d_blob_out, size_t* bytes_out);
void vt_free_device_blob(int device_id, void* d_blob);

/* ---- B streams on one GPU (one stream per camera; no cross-stream data flow) ------------ */

/* VT_PIX_YUY2: packed 4:2:2, bytes Y0 U Y1 V per pixel pair (the format the reference's IR
 * pipeline captures, src/pipeline_ir.rs:27-41, before GStreamer's videoconvert turns it into RGB);
 * converted per sampled pixel with the same BT.601 integer formulas as NV12. */
typedef enum vt_pixfmt { VT_PIX_RGB8 = 0, VT_PIX_NV12 = 1, VT_PIX_YUY2 = 2 } vt_pixfmt;

typedef struct vt_frame {        /* one device-resident frame (or a window of it) */
    const void* plane0;          /* RGB8: packed pixels; NV12: Y plane */
    const void* plane1;          /* NV12: interleaved UV plane; RGB8: NULL */
    int32_t width, height;       /* size of the FULL frame in pixels */
    int32_t stride0, stride1;    /* bytes */
    int32_t format;              /* vt_pixfmt */
    /* The planes may hold only a window of the frame: plane0 points at frame pixel
     * (origin_x, origin_y) (NV12: both even; plane1 at the matching UV pair). Pixels of the frame
     * outside the stored window must not be needed by the call (the tracker reads the search
     * window, side 4*sqrt(w*h) around the last box, plus one pixel). 0,0 = the whole frame. */
    int32_t origin_x, origin_y;
    int32_t windowed;            /* 1: the planes hold only window_w x window_h pixels (strides
                                  * describe that window); 0 with origin 0,0: the whole frame */
    /* Extent of the stored window in pixels (required when windowed == 1 or an origin is set; both
     * even for NV12, window_w even for YUY2, unless the window ends at the frame's edge). A sample
     * that falls inside the frame but outside the stored window reads as black - never out of
     * bounds. 0,0 with no origin: width x height. */
    int32_t window_w, window_h;
} vt_frame;

int vt_group_create(const char* weights_path, int device_id, const vt_config* cfg, vt_group** out);
int vt_group_create_from_device_blob(const void* d_blob, size_t bytes, int device_id,
                                     const vt_config* cfg, vt_group** out);
void vt_group_destroy(vt_group* g);
int vt_group_streams(const vt_group* g);
int vt_group_get_model_info(const vt_group* g, vt_model_info* out);
/* (re)initialise stream `stream` of the group on a device-resident frame */
int vt_group_init_device(vt_group* g, int stream, const vt_frame* frame, vt_bbox box);
/* One hot-path pass: frames[i] feeds stream i (n == vt_group_streams). Asynchronous: the pass is
 * enqueued on the group's HIP stream; results land in the group's pinned result ring. */
int vt_group_enqueue_device(vt_group* g, const vt_frame* frames, int n);
/* Wait for every enqueued pass and copy the results of the LAST pass (n entries). */
int vt_group_wait(vt_group* g, vt_result* out, int n);
/* enqueue + wait */
int vt_group_update_device(vt_group* g, const vt_frame* frames, int n, vt_result* out);
/* HIP stream the group launches on (hipStream_t as void*), for event timing by the caller */
void* vt_group_hip_stream(vt_group* g);
/* The same with HOST frames (vt_frame.plane0/plane1 are host addresses; any pixel format, strides
 * honoured, origin fields ignored): ≙ B reference hosts calling tracker.init / tracker.update
 * (src/tracker_context.rs:88,90,120) in the same frame period. Only each stream's search window is
 * read from the caller's buffers; the windows of all n frames are packed into one pinned arena and
 * cross PCIe in one copy. Synchronous: the caller's buffers may be reused on return. */
int vt_group_init_host(vt_group* g, int stream, const vt_frame* host_frame, vt_bbox box);
int vt_group_update_host(vt_group* g, const vt_frame* host_frames, int n, vt_result* out);
/* Pipelined form of the same: the upload of pass t+1 overlaps the compute of pass t.
 *   vt_group_enqueue_host(frames of t+1)   packs the windows into one of two pinned arenas and copies
 *                                          them on a separate copy stream, then enqueues the pass
 *                                          behind that copy; returns without waiting
 *   vt_group_wait_next(out)                waits for the OLDEST pass not yet collected, returns its
 *                                          results
 * At most two passes may be outstanding (one running, one queued); the host frames of an outstanding
 * pass must stay valid and unchanged until its vt_group_wait_next returns.
 * While a pass is running its boxes are not known, so the window of the next frame is cut around the
 * last KNOWN box, enlarged to cover a target that moves by a quarter of the search crop and grows by a
 * quarter in one frame (1.75x the crop side). The pixel kernel flags a pass that needed a pixel outside
 * the window it was given; vt_group_wait_next then restores the stream states from the HOST's copy of
 * the states the previous pass left (collected by the wait_next before it) and redoes that pass (and
 * the one queued behind it) with exact windows, so the results are always those of the full frames.
 * While a pass is outstanding it owns the stream states: every entry point that would advance or
 * overwrite them (vt_group_init_*, vt_group_enqueue_device, vt_group_update_device / _host,
 * vt_group_wait, vt_group_set_state_box, vt_group_profile_device) returns VT_ERR_INVALID_ARG until
 * vt_group_wait_next has collected it. (host -> tracker: src/pipeline.rs:95-101 maps the buffer on the CPU) */
int vt_group_enqueue_host(vt_group* g, const vt_frame* host_frames, int n);
int vt_group_wait_next(vt_group* g, vt_result* out, int n);
/* passes vt_group_wait_next had to redo because a speculative window missed (since creation) */
int vt_group_host_redos(const vt_group* g);
/* hipGraph captures this engine has made since creation. The pass is replayed as a captured graph, one per
 * crop-buffer tier; all of them are captured and instantiated when the engine is created (and again by
 * vt_group_set_tuning), never inside an enqueue: a live 60-fps stream (src/pipeline.rs:26-37) whose target grows
 * across a tier boundary takes no capture stall mid-track. Constant after creation unless the tuning is changed. */
int vt_group_graph_captures(const vt_group* g);

/* ---- dma-buf ingest ------------------------------------------------------------------------
 * The reference's capture side can hand out dma-bufs (v4l2src io-mode=dmabuf, src/pipeline_ir.rs:24)
 * but then maps them on the CPU (src/pipeline.rs:95-101). vt_import_dmabuf maps a dma-buf fd into
 * this device's address space (hipImportExternalMemory); *d_ptr may then be used as plane0 / plane1
 * of a vt_frame with the *_device entry points: no staging copy in host memory. The fd stays owned
 * by the caller (the library imports a dup). Whether a given exporter's buffers are importable is
 * up to the amdgpu driver; an import that the driver refuses returns VT_ERR_HIP and the caller
 * falls back to the host-pointer entry points. */
typedef struct vt_extmem vt_extmem;
int vt_import_dmabuf(int device_id, int fd, size_t bytes, vt_extmem** out, void** d_ptr);
void vt_release_dmabuf(vt_extmem* m);
/* Export the head of a hipMalloc'ed allocation as a dma-buf fd: d_ptr must be the START of the allocation (the
 * handle names the allocation, and an importer maps it from its base; a pointer inside one is refused with
 * VT_ERR_INVALID_ARG), bytes a page multiple within it; the caller closes the fd. Tooling: used by the tests to
 * exercise the import path on this machine. */
int vt_export_dmabuf(int device_id, const void* d_ptr, size_t bytes, int* fd_out);

/* ---- zero-copy ingest of host frames --------------------------------------------------------------
 * The reference maps the capture buffer on the CPU (src/pipeline.rs:95-101) and hands the tracker a view
 * of the whole frame, of which the tracker samples one window. vt_host_register page-locks such a buffer
 * (typically the capture pool, once at start-up) and maps it into the device's address space
 * (hipHostRegister + hipHostGetDevicePointer): *d_ptr + offset may then be used as plane0 / plane1 of a
 * vt_frame with the *_device entry points, and the pixel kernel reads only the pixels it samples over
 * PCIe - no staging copy, no packing on the CPU, whatever the frame size. The HOST-pointer entry points
 * (vt_init_* / vt_update_* and vt_group_*_host) recognise a frame whose planes lie inside a registered range of
 * their device and MAY take the same zero-copy route by themselves (vt_config.host_zero_copy):
 *   - single-stream engines (the reference's one tracker per process) do by default: + 1 % (1,250 -> 1,264
 *     updates/s at cfg3, 1080p), and no CPU work per frame;
 *   - batched engines do NOT by default: for them the packed-window upload is faster - 60 streams in two engines,
 *     cfg3: pipelined vt_group_enqueue_host 6,891 frames/s, synchronous vt_group_update_host 6,721, zero copy 6,369
 *     (profiles/r04_bench_cfg3_60x2_final.json) - the pixel kernel's PCIe reads are latency inside the pass, the
 *     packed upload runs beside the previous pass. host_zero_copy = 1 opts a batched engine in (a host that cannot
 *     spare the CPU time for packing: 92 % of the headline rate with no per-frame CPU work), -1 opts everything out.
 * The memory stays owned by the caller; unregister before freeing it. */
int vt_host_register(int device_id, void* host_ptr, size_t bytes, void** d_ptr);
int vt_host_unregister(int device_id, void* host_ptr);

/* ---- reference colour converter on the GPU ---------------------------------------------- */

/* ≙ nv12_full_to_rgb_parallel(nv12_data, width, height) (src/nv12_convert.rs:46-92): packed NV12
 * buffer (Y plane then interleaved UV, stride == width) -> (H,W,3) RGB8. Bit-exact with the
 * reference, including the all-zero frame when len < w*h*3/2 (src/nv12_convert.rs:48-50). For odd
 * w or h the reference reads past w*h*3/2; here len must cover those reads or the call fails with
 * VT_ERR_SHORT_BUFFER. Host pointers. */
int vt_nv12_to_rgb8(int device_id, const uint8_t* nv12, size_t len, int w, int h, uint8_t* rgb_out);
/* device-pointer form (d_rgb_out: w*h*3 bytes); enqueued on hip_stream (NULL → default) */
int vt_nv12_to_rgb8_device(int device_id, const void* d_nv12, size_t len, int w, int h,
                           void* d_rgb_out, void* hip_stream);

/* n frames per launch (a host that still wants RGB for all its cameras: the reference converts every frame,
 * src/pipeline.rs:105): frame i is d_nv12[i] (packed NV12 of lens[i] bytes, as above) -> d_rgb_out[i] (w*h*3 bytes);
 * all frames w x h. d_nv12 / lens / d_rgb_out are HOST arrays of n entries holding DEVICE pointers; they are consumed
 * before the call returns. Per frame bit-exact with vt_nv12_to_rgb8_device, including the all-zero frame for
 * lens[i] < w*h*3/2; VT_ERR_SHORT_BUFFER (nothing enqueued) if some lens[i] does not cover the conversion's reads.
 * One launch per 64 frames, enqueued on hip_stream (NULL -> default). A single 1080p conversion is a 3.9-us launch
 * bounded by its ramp-up (0.29 of the HBM roof); 30 frames in one launch stream at the rate DESIGN.md section 4 gives. */
int vt_nv12_to_rgb8_batch_device(int device_id, const void* const* d_nv12, const size_t* lens, int n, int w, int h,
                                 void* const* d_rgb_out, void* hip_stream);

/* ---- overlay drawing on the GPU (the reference's per-frame overlays) --------------------------- */

/* ≙ draw_background_nv12 / draw_text_nv12 / draw_rect_nv12 / draw_crosshair_nv12
 * (src/nv12_convert.rs:172-343) and draw_cursor / draw_selection (src/drawing.rs:5-50), applied in
 * list order to the luma plane of an NV12 frame, bit-exact with the reference's CPU loops. */
typedef enum vt_draw_type {
    VT_DRAW_BACKGROUND = 0, /* x, y, w, h; value = darkness                                */
    VT_DRAW_TEXT = 1,       /* x, y; p = scale; value = brightness; text (5x7 font, 40 glyphs) */
    VT_DRAW_RECT = 2,       /* x, y, w, h; p = thickness; value = brightness                */
    VT_DRAW_CROSSHAIR = 3,  /* x, y = centre; p = size; value = brightness                  */
    VT_DRAW_CURSOR = 4,     /* x, y                                                         */
    VT_DRAW_SELECTION = 5   /* x, y = start corner; w, h = cursor corner (dashed frame)     */
} vt_draw_type;

typedef struct vt_draw_cmd {
    int32_t type;           /* vt_draw_type */
    int32_t x, y, w, h;
    int32_t p;
    int32_t value;
    char text[36];          /* NUL-terminated, VT_DRAW_TEXT only */
} vt_draw_cmd;

/* Apply n commands to a device-resident luma plane (width x height, `stride` bytes per row),
 * enqueued on hip_stream (NULL = default stream); the command list is copied before returning. */
int vt_overlay_nv12_device(int device_id, void* d_y, int width, int height, int stride,
                           const vt_draw_cmd* cmds, int n, void* hip_stream);
/* Host-pointer form: draws into the packed NV12 buffer (stride == width) in place. */
int vt_overlay_nv12(int device_id, uint8_t* nv12, int width, int height, const vt_draw_cmd* cmds,
                    int n);
/* The packed-RGB8 variants the reference's live pipeline uses (src/drawing_rgb.rs:30-129,
 * src/pipeline_ir.rs:168-202): same command list; value = 0xRRGGBB for rect / crosshair, luma for
 * text; background fills with 30, cursor is (0,255,0), selection (255,255,0) as in the reference. */
int vt_overlay_rgb8_device(int device_id, void* d_rgb, int width, int height, int stride,
                           const vt_draw_cmd* cmds, int n, void* hip_stream);
int vt_overlay_rgb8(int device_id, uint8_t* rgb, int width, int height, const vt_draw_cmd* cmds,
                    int n);

/* ---- per-kernel timing and stage taps (parity tests, bench roofline) -------------------- */

typedef struct vt_kernel_time {
    char name[48];       /* kernel family, e.g. "gemm_bf16_resid" */
    int32_t launches;    /* launches of that family in one pass */
    float ms_total;      /* summed HIP-event time of those launches in one pass */
    double flops;        /* algorithmic FLOPs of those launches (0 for byte-bound kernels) */
    double bytes;        /* algorithmic bytes of those launches */
} vt_kernel_time;

/* Run `iters` eager passes over `frames` with HIP events around every launch (events on the
 * group's own stream) and return per-family averages per pass. Advances tracker state like
 * `iters` updates. Returns the number of families written (<= max_out) or a negative vt_status. */
int vt_group_profile_device(vt_group* g, const vt_frame* frames, int n, int iters,
                            vt_kernel_time* out, int max_out);

/* Stage taps: when enabled the pass runs eagerly and keeps a copy of the residual stream after the
 * patch embedding and after every encoder block (for stage-level parity tests). */
int vt_group_enable_taps(vt_group* g, int enable);
/* Diagnostics (A/B measurements and parity tests of alternative kernels; results are the same quantity either
 * way): key "head_band": 2 (default; any negative value selects it) = the head's convolutions on the band kernel, the
 * final LayerNorm inside the first layer's launch and the logits and the decode behind the last layer's, 1 = the same
 * with the LayerNorm as a launch of its own, 0 = implicit GEMMs + head_out + decode launches; key "crop_tier": >= 0 forces the crop
 * kernel's LDS buffer tier (0: 16 KiB, 1: 32 KiB, 2: 64 KiB), < 0 (default) = chosen per pass from the boxes the host
 * knows. Not while a pipelined pass is outstanding. */
int vt_group_set_tuning(vt_group* g, const char* key, int value);
/* A single tracker viewed as a group of one (taps, profiling, stream handle). The view belongs to
 * the tracker: valid until vt_destroy(t), the same pointer on every call, never to be destroyed
 * by the caller. */
vt_group* vt_tracker_as_group(vt_tracker* t);

/* Overwrite the box the next update of `stream` crops its search window around (x, y, w, h in frame
 * pixels). Tooling hook: lets a caller evaluate the network on a window of its choosing (head
 * training data, stage tests) while keeping the template set by vt_group_init_device. */
int vt_group_set_state_box(vt_group* g, int stream, const float* box4);

/* Copy an intermediate tensor of the last pass to the host as float32.
 * names: "patches" [N,Kpad], "tokens0" [N,D], "layer<i>" [N,D] (residual stream after block i;
 * both need taps), "x" [N,D] (final residual stream; like the taps the value of the 3-byte pair it is stored
 * as: bf16 + a signed byte in units of 2^-12), "rowstat" [N,2] (row terms of the last folded LayerNorm), "attn" [N,D] (last block's attention output),
 * "feat" [Ns,D], "head_t3" [Ns,C], "head_out" [Ns,8] (score,ox,oy,w,h logits),
 * "state" (the stream's device state record as raw 32-bit words), "graph_replays" [3] (passes replayed so far
 * per crop-buffer tier: which of the captured graphs ran).
 * Returns the element count, or a negative vt_status. With out == NULL only the count. */
int64_t vt_group_read_tensor(vt_group* g, int stream, const char* name, float* out,
                             int64_t capacity);

#ifdef __cplusplus
}
#endif
#endif /* VITTRACK_HIP_H */
