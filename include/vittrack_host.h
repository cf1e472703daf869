/*
 * vittrack_host.h — C ABI of libvittrack_host.so: the C++ mirror of the reference's host-side
 * control layer around the tracker, exported so that tests and foreign-language hosts can drive
 * it. The reference host itself is Rust and keeps its own copies of these types; this library
 * exists because cargo/rustc are not in the build image and the per-frame call
 * sequence still has to be exercised end to end.
 *
 * Mirrors (same names, argument meaning, thresholds):
 *   TrackerContext::{new, handle_command, process_frame, state_name}  src/tracker_context.rs:19,36,64,157
 *   SelectionState::{new, move_cursor, get_bbox}                      src/selection_state.rs:21,33,39
 *   TimingStats::{add_interval, add_times, fps, avg_conv_ms, avg_track_ms}  src/timing_stats.rs:18-60
 *   the per-frame probe closure (minus drawing)                       src/pipeline.rs:67-184
 */
#ifndef VITTRACK_HOST_H
#define VITTRACK_HOST_H

#include <stdint.h>

#include "vittrack_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ≙ enum UserCommand (src/user_commands.rs:1-10) */
typedef enum vth_command {
    VTH_MOVE_UP = 0, VTH_MOVE_DOWN = 1, VTH_MOVE_LEFT = 2, VTH_MOVE_RIGHT = 3,
    VTH_CONFIRM = 4, VTH_CANCEL = 5, VTH_QUIT = 6
} vth_command;

/* ≙ enum AppState (src/app_state.rs:1-6) */
typedef enum vth_state_kind { VTH_SELECTING = 0, VTH_TRACKING = 1, VTH_LOST = 2 } vth_state_kind;

typedef struct vth_ctx vth_ctx;       /* ≙ TrackerContext */
typedef struct vth_timing vth_timing; /* ≙ TimingStats */

/* A tracker supplied by the caller instead of VitTrack (scripted scores in tests, other back
 * ends). update returns 0 and fills *out for Ok(result), non-zero for the reference's Err arm. */
typedef struct vth_tracker_callbacks {
    void (*init)(void* user, const uint8_t* data, int w, int h, int stride, int format, vt_bbox box);
    int (*update)(void* user, const uint8_t* data, int w, int h, int stride, int format,
                  vt_result* out);
} vth_tracker_callbacks;

typedef struct vth_ctx_info {
    int32_t state_kind;        /* vth_state_kind */
    uint64_t lost_frames;      /* Lost { frames } */
    int32_t has_bbox;          /* current_bbox.is_some() */
    vt_bbox current_bbox;
    float current_score;
    int32_t pending_confirm;
    int32_t cursor_x, cursor_y, start_x, start_y;
    int32_t selection_phase;   /* 0 MovingToStart, 1 SelectingArea */
    int32_t frame_width, frame_height;
} vth_ctx_info;

const char* vth_last_error(void);

/* ≙ TrackerContext::new(model_path, width, height): loads the model through VitTrack::new on
 * `device`. Returns 0 or a negative vt_status (e.g. VT_ERR_NO_DEVICE; text: "Failed: ..."). */
int vth_ctx_new(const char* model_path, int width, int height, int device, vth_ctx** out);
int vth_ctx_new_with_tracker(vth_tracker_callbacks cb, void* user, int width, int height,
                             vth_ctx** out);
void vth_ctx_free(vth_ctx* c);
/* ≙ handle_command(UserCommand); `fast` only matters for the four moves */
void vth_ctx_handle_command(vth_ctx* c, int command, int fast);
/* ≙ process_frame(&full_image) -> Option<BBox>: returns 1 and fills *out for Some, 0 for None */
int vth_ctx_process_frame_rgb8(vth_ctx* c, const uint8_t* rgb, int w, int h, int stride,
                               vt_bbox* out);
/* fused ingest: the packed NV12 buffer as GStreamer maps it (Y then UV, stride == width) */
int vth_ctx_process_frame_nv12(vth_ctx* c, const uint8_t* nv12, int w, int h, vt_bbox* out);
const char* vth_ctx_state_name(const vth_ctx* c);
void vth_ctx_get_info(const vth_ctx* c, vth_ctx_info* out);

/* ≙ SelectionState::get_bbox on explicit corners (src/selection_state.rs:39-45) */
vt_bbox vth_selection_bbox(int start_x, int start_y, int cursor_x, int cursor_y);

vth_timing* vth_timing_new(void);
void vth_timing_free(vth_timing* t);
void vth_timing_add_interval(vth_timing* t, uint64_t us);
void vth_timing_add_times(vth_timing* t, uint64_t conv_us, uint64_t track_us);
double vth_timing_fps(const vth_timing* t);
double vth_timing_avg_conv_ms(const vth_timing* t);
double vth_timing_avg_track_ms(const vth_timing* t);

/* ≙ nv12_full_to_rgb_parallel (src/nv12_convert.rs:46) on the GPU, through libvittrack_hip.so */
int vth_nv12_full_to_rgb(int device, const uint8_t* nv12, size_t len, int w, int h, uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
