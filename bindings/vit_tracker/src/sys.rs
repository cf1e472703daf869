//! Raw declarations of include/vittrack_hip.h (VT_ABI_VERSION 5), one for one and in header order.
//! tests/test_rust_binding.py parses this file and the header and fails on any drift: a missing or
//! extra function, a different argument count / order / type, a struct whose fields or size differ.
#![allow(non_camel_case_types, dead_code)]

use std::ffi::{c_char, c_int, c_void};

pub const VT_ABI_VERSION: c_int = 5;
pub const VT_MAX_STREAMS: c_int = 1024;
pub const VT_RCCL_ID_BYTES: usize = 128;

// vt_status
pub const VT_OK: c_int = 0;
pub const VT_ERR_INVALID_ARG: c_int = -1;
pub const VT_ERR_NO_DEVICE: c_int = -2;
pub const VT_ERR_IO: c_int = -3;
pub const VT_ERR_FORMAT: c_int = -4;
pub const VT_ERR_HIP: c_int = -5;
pub const VT_ERR_NOT_INITIALIZED: c_int = -6;
pub const VT_ERR_SHORT_BUFFER: c_int = -7;
pub const VT_ERR_OOM: c_int = -8;

// vt_pixfmt
pub const VT_PIX_RGB8: i32 = 0;
pub const VT_PIX_NV12: i32 = 1;
pub const VT_PIX_YUY2: i32 = 2;

/// ≙ vt_bbox ≙ vit_tracker::BBox (src/selection_state.rs:44, src/tracker_context.rs:85)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct BBox {
    pub x: i32,
    pub y: i32,
    pub width: i32,
    pub height: i32,
}

/// ≙ vt_result (fields used at src/tracker_context.rs:92-95,122-125)
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct VtResult {
    pub success: i32,
    pub score: f32,
    pub bbox: BBox,
}

/// ≙ vt_config
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct VtConfig {
    pub struct_size: u32,
    pub success_threshold: f32,
    pub use_graph: i32,
    pub n_streams: i32,
    pub max_frame_width: i32,
    pub max_frame_height: i32,
    pub max_device_mib: i32,
    pub host_window_margin_pct: i32,
    pub host_zero_copy: i32,
    pub reserved: [i32; 5],
}

/// ≙ vt_model_info
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct VtModelInfo {
    pub patch: i32,
    pub template_size: i32,
    pub search_size: i32,
    pub dim: i32,
    pub heads: i32,
    pub layers: i32,
    pub mlp_dim: i32,
    pub head_channels: i32,
    pub tokens_template: i32,
    pub tokens_search: i32,
    pub kpad: i32,
    pub score_grid: i32,
    pub flops_per_frame: f64,
    pub encoder_flops_per_frame: f64,
    pub weight_bytes: u64,
}

/// ≙ vt_frame
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct VtFrame {
    pub plane0: *const c_void,
    pub plane1: *const c_void,
    pub width: i32,
    pub height: i32,
    pub stride0: i32,
    pub stride1: i32,
    pub format: i32,
    pub origin_x: i32,
    pub origin_y: i32,
    pub windowed: i32,
    pub window_w: i32,
    pub window_h: i32,
}

/// ≙ vt_draw_cmd
#[repr(C)]
#[derive(Clone, Copy)]
pub struct VtDrawCmd {
    pub r#type: i32,
    pub x: i32,
    pub y: i32,
    pub w: i32,
    pub h: i32,
    pub p: i32,
    pub value: i32,
    pub text: [c_char; 36],
}

/// ≙ vt_kernel_time
#[repr(C)]
#[derive(Clone, Copy)]
pub struct VtKernelTime {
    pub name: [c_char; 48],
    pub launches: i32,
    pub ms_total: f32,
    pub flops: f64,
    pub bytes: f64,
}

#[repr(C)]
pub struct vt_tracker {
    _private: [u8; 0],
}
#[repr(C)]
pub struct vt_group {
    _private: [u8; 0],
}
#[repr(C)]
pub struct vt_extmem {
    _private: [u8; 0],
}

extern "C" {
    pub fn vt_config_default(cfg: *mut VtConfig);
    pub fn vt_last_error() -> *const c_char;
    pub fn vt_abi_version() -> c_int;
    pub fn vt_build_info() -> *const c_char;
    pub fn vt_device_count() -> c_int;
    pub fn vt_recommended_streams(info: *const VtModelInfo, max_streams: c_int) -> c_int;
    pub fn vt_plan_engines(info: *const VtModelInfo, n_streams: c_int, sizes: *mut c_int, cap: c_int) -> c_int;

    pub fn vt_create(weights_path: *const c_char, device_id: c_int, cfg: *const VtConfig, out: *mut *mut vt_tracker) -> c_int;
    pub fn vt_create_from_device_blob(d_blob: *const c_void, bytes: usize, device_id: c_int, cfg: *const VtConfig, out: *mut *mut vt_tracker) -> c_int;
    pub fn vt_destroy(t: *mut vt_tracker);
    pub fn vt_get_model_info(t: *const vt_tracker, out: *mut VtModelInfo) -> c_int;

    pub fn vt_init_rgb8(t: *mut vt_tracker, rgb: *const u8, w: c_int, h: c_int, stride_bytes: c_int, bbox: BBox) -> c_int;
    pub fn vt_update_rgb8(t: *mut vt_tracker, rgb: *const u8, w: c_int, h: c_int, stride_bytes: c_int, out: *mut VtResult) -> c_int;
    pub fn vt_init_nv12(t: *mut vt_tracker, y: *const u8, uv: *const u8, w: c_int, h: c_int, y_stride: c_int, uv_stride: c_int, bbox: BBox) -> c_int;
    pub fn vt_update_nv12(t: *mut vt_tracker, y: *const u8, uv: *const u8, w: c_int, h: c_int, y_stride: c_int, uv_stride: c_int, out: *mut VtResult) -> c_int;
    pub fn vt_init_yuy2(t: *mut vt_tracker, yuy2: *const u8, w: c_int, h: c_int, stride_bytes: c_int, bbox: BBox) -> c_int;
    pub fn vt_update_yuy2(t: *mut vt_tracker, yuy2: *const u8, w: c_int, h: c_int, stride_bytes: c_int, out: *mut VtResult) -> c_int;

    pub fn vt_init_rgb8_device(t: *mut vt_tracker, d_rgb: *const c_void, w: c_int, h: c_int, stride_bytes: c_int, bbox: BBox) -> c_int;
    pub fn vt_update_rgb8_device(t: *mut vt_tracker, d_rgb: *const c_void, w: c_int, h: c_int, stride_bytes: c_int, out: *mut VtResult) -> c_int;
    pub fn vt_init_nv12_device(t: *mut vt_tracker, d_y: *const c_void, d_uv: *const c_void, w: c_int, h: c_int, y_stride: c_int, uv_stride: c_int, bbox: BBox) -> c_int;
    pub fn vt_update_nv12_device(t: *mut vt_tracker, d_y: *const c_void, d_uv: *const c_void, w: c_int, h: c_int, y_stride: c_int, uv_stride: c_int, out: *mut VtResult) -> c_int;

    pub fn vt_rccl_unique_id(id_out: *mut u8) -> c_int;
    pub fn vt_broadcast_weights_rccl(id: *const u8, world: c_int, rank: c_int, device_id: c_int, weights_path: *const c_char, d_blob_out: *mut *mut c_void, bytes_out: *mut usize) -> c_int;
    pub fn vt_free_device_blob(device_id: c_int, d_blob: *mut c_void);

    pub fn vt_group_create(weights_path: *const c_char, device_id: c_int, cfg: *const VtConfig, out: *mut *mut vt_group) -> c_int;
    pub fn vt_group_create_from_device_blob(d_blob: *const c_void, bytes: usize, device_id: c_int, cfg: *const VtConfig, out: *mut *mut vt_group) -> c_int;
    pub fn vt_group_destroy(g: *mut vt_group);
    pub fn vt_group_streams(g: *const vt_group) -> c_int;
    pub fn vt_group_get_model_info(g: *const vt_group, out: *mut VtModelInfo) -> c_int;
    pub fn vt_group_init_device(g: *mut vt_group, stream: c_int, frame: *const VtFrame, bbox: BBox) -> c_int;
    pub fn vt_group_enqueue_device(g: *mut vt_group, frames: *const VtFrame, n: c_int) -> c_int;
    pub fn vt_group_wait(g: *mut vt_group, out: *mut VtResult, n: c_int) -> c_int;
    pub fn vt_group_update_device(g: *mut vt_group, frames: *const VtFrame, n: c_int, out: *mut VtResult) -> c_int;
    pub fn vt_group_hip_stream(g: *mut vt_group) -> *mut c_void;
    pub fn vt_group_init_host(g: *mut vt_group, stream: c_int, host_frame: *const VtFrame, bbox: BBox) -> c_int;
    pub fn vt_group_update_host(g: *mut vt_group, host_frames: *const VtFrame, n: c_int, out: *mut VtResult) -> c_int;
    pub fn vt_group_enqueue_host(g: *mut vt_group, host_frames: *const VtFrame, n: c_int) -> c_int;
    pub fn vt_group_wait_next(g: *mut vt_group, out: *mut VtResult, n: c_int) -> c_int;
    pub fn vt_group_host_redos(g: *const vt_group) -> c_int;
    pub fn vt_group_graph_captures(g: *const vt_group) -> c_int;

    pub fn vt_import_dmabuf(device_id: c_int, fd: c_int, bytes: usize, out: *mut *mut vt_extmem, d_ptr: *mut *mut c_void) -> c_int;
    pub fn vt_release_dmabuf(m: *mut vt_extmem);
    pub fn vt_export_dmabuf(device_id: c_int, d_ptr: *const c_void, bytes: usize, fd_out: *mut c_int) -> c_int;

    pub fn vt_host_register(device_id: c_int, host_ptr: *mut c_void, bytes: usize, d_ptr: *mut *mut c_void) -> c_int;
    pub fn vt_host_unregister(device_id: c_int, host_ptr: *mut c_void) -> c_int;

    pub fn vt_nv12_to_rgb8(device_id: c_int, nv12: *const u8, len: usize, w: c_int, h: c_int, rgb_out: *mut u8) -> c_int;
    pub fn vt_nv12_to_rgb8_batch_device(device_id: c_int, d_nv12: *const *const c_void, lens: *const usize, n: c_int, w: c_int, h: c_int, d_rgb_out: *const *mut c_void, hip_stream: *mut c_void) -> c_int;
    pub fn vt_nv12_to_rgb8_device(device_id: c_int, d_nv12: *const c_void, len: usize, w: c_int, h: c_int, d_rgb_out: *mut c_void, hip_stream: *mut c_void) -> c_int;

    pub fn vt_overlay_nv12_device(device_id: c_int, d_y: *mut c_void, width: c_int, height: c_int, stride: c_int, cmds: *const VtDrawCmd, n: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn vt_overlay_nv12(device_id: c_int, nv12: *mut u8, width: c_int, height: c_int, cmds: *const VtDrawCmd, n: c_int) -> c_int;
    pub fn vt_overlay_rgb8_device(device_id: c_int, d_rgb: *mut c_void, width: c_int, height: c_int, stride: c_int, cmds: *const VtDrawCmd, n: c_int, hip_stream: *mut c_void) -> c_int;
    pub fn vt_overlay_rgb8(device_id: c_int, rgb: *mut u8, width: c_int, height: c_int, cmds: *const VtDrawCmd, n: c_int) -> c_int;

    pub fn vt_group_profile_device(g: *mut vt_group, frames: *const VtFrame, n: c_int, iters: c_int, out: *mut VtKernelTime, max_out: c_int) -> c_int;
    pub fn vt_group_enable_taps(g: *mut vt_group, enable: c_int) -> c_int;
    pub fn vt_group_set_tuning(g: *mut vt_group, key: *const c_char, value: c_int) -> c_int;
    pub fn vt_tracker_as_group(t: *mut vt_tracker) -> *mut vt_group;
    pub fn vt_group_set_state_box(g: *mut vt_group, stream: c_int, box4: *const f32) -> c_int;
    pub fn vt_group_read_tensor(g: *mut vt_group, stream: c_int, name: *const c_char, out: *mut f32, capacity: i64) -> i64;

}
