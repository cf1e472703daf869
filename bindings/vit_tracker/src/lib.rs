//! `vit_tracker` over libvittrack_hip.so: the API the reference host links today
//! (`vit_tracker::{VitTrack, BBox}`, /root/reference/src/tracker_context.rs:2,
//! src/selection_state.rs:1), unchanged in names, argument meaning and error behaviour, so that
//! tracker_context.rs, selection_state.rs, pipeline*.rs and nv12_convert.rs compile as they are.
//!
//!   VitTrack::new(model_path)      src/tracker_context.rs:21     -> vt_create
//!   tracker.init(&view, bbox)      src/tracker_context.rs:88     -> vt_init_rgb8   (result ignored by the host)
//!   tracker.update(&view)          src/tracker_context.rs:90,120 -> vt_update_rgb8
//!   BBox::new / BBox::from_array   src/selection_state.rs:44, src/tracker_context.rs:94
//!
//! Nothing unwinds across the C boundary (the host is built with panic = "abort", Cargo.toml:37):
//! every C entry returns a status code, surfaced here as `Err(TrackError)`.
pub mod sys;

use ndarray::ArrayView3;
use std::ffi::{c_int, CStr, CString};

pub use sys::BBox;

impl BBox {
    /// src/selection_state.rs:44
    pub fn new(x: i32, y: i32, width: i32, height: i32) -> Self {
        Self { x, y, width, height }
    }
    /// src/tracker_context.rs:94,123
    pub fn from_array(a: &[i32; 4]) -> Self {
        Self::new(a[0], a[1], a[2], a[3])
    }
}

/// Error of a library call: the vt_status code and vt_last_error()'s text. `Debug` is what the host
/// prints (`{:?}`, src/tracker_context.rs:22,106,135).
#[derive(Debug, Clone)]
pub struct TrackError {
    pub code: i32,
    pub text: String,
}
impl std::fmt::Display for TrackError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "vittrack_hip error {}: {}", self.code, self.text)
    }
}
impl std::error::Error for TrackError {}

fn last(code: c_int) -> TrackError {
    let text = unsafe { CStr::from_ptr(sys::vt_last_error()) }.to_string_lossy().into_owned();
    TrackError { code, text }
}

/// What `update` returns: the fields the host reads at src/tracker_context.rs:92-95,122-125.
#[derive(Debug, Clone, Copy)]
pub struct TrackResult {
    pub success: bool,
    pub score: f32,
    pub bbox: [i32; 4],
}
impl From<sys::VtResult> for TrackResult {
    fn from(r: sys::VtResult) -> Self {
        Self { success: r.success != 0, score: r.score, bbox: [r.bbox.x, r.bbox.y, r.bbox.width, r.bbox.height] }
    }
}

pub struct VitTrack {
    h: *mut sys::vt_tracker,
}
// Constructed on the main thread (src/main.rs:49 -> src/pipeline_ir.rs:89), used only on the GStreamer
// streaming thread behind a Mutex (src/pipeline.rs:55-67,110-119). The C handle has no thread affinity
// (every entry point selects and restores the HIP device): tests/test_gpu_threading.py.
unsafe impl Send for VitTrack {}

impl VitTrack {
    /// ≙ src/tracker_context.rs:21. `model_path` is a VTWB0001 weight blob; the device comes from
    /// VITTRACK_DEVICE (default 0).
    pub fn new(model_path: &str) -> Result<Self, TrackError> {
        let dev = std::env::var("VITTRACK_DEVICE").ok().and_then(|s| s.parse().ok()).unwrap_or(0);
        Self::with_device(model_path, dev)
    }

    pub fn with_device(model_path: &str, device: i32) -> Result<Self, TrackError> {
        let p = CString::new(model_path).map_err(|_| TrackError { code: sys::VT_ERR_INVALID_ARG, text: "path has NUL".into() })?;
        let mut cfg = std::mem::MaybeUninit::<sys::VtConfig>::uninit();
        let mut h = std::ptr::null_mut();
        let rc = unsafe {
            sys::vt_config_default(cfg.as_mut_ptr());
            sys::vt_create(p.as_ptr(), device, cfg.as_ptr(), &mut h)
        };
        if rc != sys::VT_OK {
            Err(last(rc))
        } else {
            Ok(Self { h })
        }
    }

    fn rgb_view(img: &ArrayView3<u8>) -> (*const u8, c_int, c_int, c_int) {
        let (h, w, c) = img.dim();
        let s = img.strides(); // (W*3, 3, 1) for the views the host builds (src/nv12_convert.rs:90, src/pipeline_ir.rs:142)
        assert!(c == 3 && s[2] == 1 && s[1] == 3, "RGB8 HWC view expected");
        (img.as_ptr(), w as c_int, h as c_int, s[0] as c_int)
    }

    /// ≙ src/tracker_context.rs:88 (the host discards the result; `()` keeps its code unchanged)
    pub fn init(&mut self, img: &ArrayView3<u8>, bbox: BBox) {
        let (p, w, h, s) = Self::rgb_view(img);
        unsafe {
            sys::vt_init_rgb8(self.h, p, w, h, s, bbox);
        }
    }

    /// ≙ src/tracker_context.rs:90,120
    pub fn update(&mut self, img: &ArrayView3<u8>) -> Result<TrackResult, TrackError> {
        let (p, w, h, s) = Self::rgb_view(img);
        let mut r = sys::VtResult::default();
        let rc = unsafe { sys::vt_update_rgb8(self.h, p, w, h, s, &mut r) };
        if rc != sys::VT_OK {
            return Err(last(rc));
        }
        Ok(r.into())
    }

    /// Fused NV12 ingest (not in the original crate): the same result as init/update on the RGB frame
    /// nv12_full_to_rgb_parallel (src/nv12_convert.rs:46) would have produced, without converting the
    /// whole frame; lets src/pipeline.rs:104-106 go. `nv12` is the mapped buffer (Y plane then
    /// interleaved UV, stride == width as src/nv12_convert.rs:53-54 assumes).
    pub fn init_nv12(&mut self, nv12: &[u8], w: usize, h: usize, bbox: BBox) {
        if nv12.len() < w * h * 3 / 2 {
            return;
        }
        unsafe {
            sys::vt_init_nv12(self.h, nv12.as_ptr(), nv12[w * h..].as_ptr(), w as c_int, h as c_int, w as c_int, w as c_int, bbox);
        }
    }

    pub fn update_nv12(&mut self, nv12: &[u8], w: usize, h: usize) -> Result<TrackResult, TrackError> {
        if nv12.len() < w * h * 3 / 2 {
            return Err(TrackError { code: sys::VT_ERR_SHORT_BUFFER, text: "nv12 buffer shorter than w*h*3/2".into() });
        }
        let mut r = sys::VtResult::default();
        let rc = unsafe {
            sys::vt_update_nv12(self.h, nv12.as_ptr(), nv12[w * h..].as_ptr(), w as c_int, h as c_int, w as c_int, w as c_int, &mut r)
        };
        if rc != sys::VT_OK {
            return Err(last(rc));
        }
        Ok(r.into())
    }

    /// Fused YUY2 ingest: the capture format of the live IR pipeline (src/pipeline_ir.rs:27-41)
    pub fn update_yuy2(&mut self, yuy2: &[u8], w: usize, h: usize) -> Result<TrackResult, TrackError> {
        if yuy2.len() < w * h * 2 {
            return Err(TrackError { code: sys::VT_ERR_SHORT_BUFFER, text: "yuy2 buffer shorter than w*h*2".into() });
        }
        let mut r = sys::VtResult::default();
        let rc = unsafe { sys::vt_update_yuy2(self.h, yuy2.as_ptr(), w as c_int, h as c_int, (2 * w) as c_int, &mut r) };
        if rc != sys::VT_OK {
            return Err(last(rc));
        }
        Ok(r.into())
    }

    pub fn model_info(&self) -> Result<sys::VtModelInfo, TrackError> {
        let mut mi = sys::VtModelInfo::default();
        let rc = unsafe { sys::vt_get_model_info(self.h, &mut mi) };
        if rc != sys::VT_OK {
            return Err(last(rc));
        }
        Ok(mi)
    }
}

impl Drop for VitTrack {
    fn drop(&mut self) {
        unsafe { sys::vt_destroy(self.h) }
    }
}

/// ≙ nv12_full_to_rgb_parallel (src/nv12_convert.rs:46-92) on the GPU, bit for bit (including the
/// all-zero frame for a short buffer, :48-50); for callers that still want the whole RGB frame.
pub fn nv12_full_to_rgb(nv12: &[u8], w: usize, h: usize, device: i32) -> Result<Vec<u8>, TrackError> {
    let mut out = vec![0u8; w * h * 3];
    let rc = unsafe { sys::vt_nv12_to_rgb8(device, nv12.as_ptr(), nv12.len(), w as c_int, h as c_int, out.as_mut_ptr()) };
    if rc != sys::VT_OK {
        return Err(last(rc));
    }
    Ok(out)
}
