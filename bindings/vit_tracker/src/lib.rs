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
//! every C entry returns a status code, surfaced here as `Err(TrackError)`. For the same reason this
//! file contains no `assert!`, `unwrap()`, `expect(` or `panic!` and no slice indexing that can fail
//! (tests/test_rust_binding.py greps for them): a view the library cannot take is an `Err` from
//! `update`; `init`, whose result the host discards (src/tracker_context.rs:88), records the error and
//! the next `update` returns it.
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
    /// an `init` that failed (the host ignores init's result): returned by the next `update`
    pending: Option<TrackError>,
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
            Ok(Self { h, pending: None })
        }
    }

    /// The (H, W, 3) RGB8 views the host builds (src/nv12_convert.rs:90, src/pipeline_ir.rs:142) have strides
    /// (W*3, 3, 1); rows may be padded. Anything else is VT_ERR_INVALID_ARG - never a panic (panic = "abort").
    fn rgb_view(img: &ArrayView3<u8>) -> Result<(*const u8, c_int, c_int, c_int), TrackError> {
        let (h, w, c) = img.dim();
        let bad = |text: &str| TrackError { code: sys::VT_ERR_INVALID_ARG, text: text.into() };
        let (s0, s1, s2) = match img.strides() {
            [a, b, c] => (*a, *b, *c),
            _ => return Err(bad("3-dimensional view expected")),
        };
        if c != 3 || s2 != 1 || s1 != 3 || s0 < 3 * w as isize {
            return Err(bad("RGB8 HWC view with strides (>= W*3, 3, 1) expected"));
        }
        if w > c_int::MAX as usize || h > c_int::MAX as usize || s0 > c_int::MAX as isize {
            return Err(bad("view too large"));
        }
        Ok((img.as_ptr(), w as c_int, h as c_int, s0 as c_int))
    }

    /// ≙ src/tracker_context.rs:88 (the host discards the result; `()` keeps its code unchanged). A failure is kept
    /// and returned by the next `update` (which the host calls on the same frame, :90).
    pub fn init(&mut self, img: &ArrayView3<u8>, bbox: BBox) {
        self.pending = match Self::rgb_view(img) {
            Err(e) => Some(e),
            Ok((p, w, h, s)) => {
                let rc = unsafe { sys::vt_init_rgb8(self.h, p, w, h, s, bbox) };
                if rc != sys::VT_OK { Some(last(rc)) } else { None }
            }
        };
    }

    /// ≙ src/tracker_context.rs:90,120
    pub fn update(&mut self, img: &ArrayView3<u8>) -> Result<TrackResult, TrackError> {
        if let Some(e) = self.pending.take() {
            return Err(e);
        }
        let (p, w, h, s) = Self::rgb_view(img)?;
        let mut r = sys::VtResult::default();
        let rc = unsafe { sys::vt_update_rgb8(self.h, p, w, h, s, &mut r) };
        if rc != sys::VT_OK {
            return Err(last(rc));
        }
        Ok(r.into())
    }

    /// (Y plane, UV plane) of a packed NV12 buffer, or VT_ERR_SHORT_BUFFER / VT_ERR_INVALID_ARG - no indexing that can panic
    fn nv12_planes(nv12: &[u8], w: usize, h: usize) -> Result<(*const u8, *const u8, c_int, c_int), TrackError> {
        let px = w.checked_mul(h).filter(|_| w <= c_int::MAX as usize && h <= c_int::MAX as usize);
        let px = px.ok_or(TrackError { code: sys::VT_ERR_INVALID_ARG, text: "frame size out of range".into() })?;
        let uv = nv12.get(px..).filter(|uv| uv.len() >= px / 2);
        let uv = uv.ok_or(TrackError { code: sys::VT_ERR_SHORT_BUFFER, text: "nv12 buffer shorter than w*h*3/2".into() })?;
        Ok((nv12.as_ptr(), uv.as_ptr(), w as c_int, h as c_int))
    }

    /// Fused NV12 ingest (not in the original crate): the same result as init/update on the RGB frame
    /// nv12_full_to_rgb_parallel (src/nv12_convert.rs:46) would have produced, without converting the
    /// whole frame; lets src/pipeline.rs:104-106 go. `nv12` is the mapped buffer (Y plane then
    /// interleaved UV, stride == width as src/nv12_convert.rs:53-54 assumes).
    pub fn init_nv12(&mut self, nv12: &[u8], w: usize, h: usize, bbox: BBox) {
        self.pending = match Self::nv12_planes(nv12, w, h) {
            Err(e) => Some(e),
            Ok((y, uv, wi, hi)) => {
                let rc = unsafe { sys::vt_init_nv12(self.h, y, uv, wi, hi, wi, wi, bbox) };
                if rc != sys::VT_OK { Some(last(rc)) } else { None }
            }
        };
    }

    pub fn update_nv12(&mut self, nv12: &[u8], w: usize, h: usize) -> Result<TrackResult, TrackError> {
        if let Some(e) = self.pending.take() {
            return Err(e);
        }
        let (y, uv, wi, hi) = Self::nv12_planes(nv12, w, h)?;
        let mut r = sys::VtResult::default();
        let rc = unsafe { sys::vt_update_nv12(self.h, y, uv, wi, hi, wi, wi, &mut r) };
        if rc != sys::VT_OK {
            return Err(last(rc));
        }
        Ok(r.into())
    }

    /// Fused YUY2 ingest: the capture format of the live IR pipeline (src/pipeline_ir.rs:27-41)
    pub fn update_yuy2(&mut self, yuy2: &[u8], w: usize, h: usize) -> Result<TrackResult, TrackError> {
        if let Some(e) = self.pending.take() {
            return Err(e);
        }
        let need = w.checked_mul(h).and_then(|p| p.checked_mul(2)).filter(|_| w <= (c_int::MAX / 2) as usize && h <= c_int::MAX as usize);
        match need {
            None => return Err(TrackError { code: sys::VT_ERR_INVALID_ARG, text: "frame size out of range".into() }),
            Some(n) if yuy2.len() < n => return Err(TrackError { code: sys::VT_ERR_SHORT_BUFFER, text: "yuy2 buffer shorter than w*h*2".into() }),
            Some(_) => {}
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
    let bytes = w.checked_mul(h).and_then(|p| p.checked_mul(3)).filter(|_| w <= c_int::MAX as usize && h <= c_int::MAX as usize);
    let bytes = bytes.ok_or(TrackError { code: sys::VT_ERR_INVALID_ARG, text: "frame size out of range".into() })?;
    let mut out = vec![0u8; bytes];
    let rc = unsafe { sys::vt_nv12_to_rgb8(device, nv12.as_ptr(), nv12.len(), w as c_int, h as c_int, out.as_mut_ptr()) };
    if rc != sys::VT_OK {
        return Err(last(rc));
    }
    Ok(out)
}
