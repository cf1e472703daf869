"""MI355X-native ViT tracker hot path — Python bindings over the C ABI (include/vittrack_hip.h).

The classes mirror the reference's `vit_tracker` crate surface as its host uses it
(/root/reference/src/tracker_context.rs:2,21,88,90,94,120; src/selection_state.rs:1,44):
`VitTrack.new / init / update`, `BBox.new / from_array`. They are thin ctypes wrappers: all
computation happens in libvittrack_hip.so (hand-written gfx950 kernels). There is no CPU
fallback — without the built library or without a gfx950 device every call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import (POINTER, Structure, byref, c_char, c_char_p, c_double, c_float, c_int,
                    c_int32, c_int64, c_size_t, c_uint8, c_uint16, c_uint32, c_uint64, c_void_p)

import numpy as np

from . import weights  # noqa: F401  (blob writer / configs)
from . import synth    # noqa: F401

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VITTRACK_HIP_LIB", os.path.join(PKG_DIR, "libvittrack_hip.so"))
# operator-level entry points (include/vittrack_hip_ops.h): a library of their own, for tests and tuning tools; a tuning
# build named by VITTRACK_HIP_LIB carries them too
OPS_LIB_PATH = os.environ.get("VITTRACK_HIP_OPS_LIB") or os.environ.get("VITTRACK_HIP_LIB") or \
    os.path.join(PKG_DIR, "libvittrack_hip_ops.so")

PIX_RGB8, PIX_NV12, PIX_YUY2 = 0, 1, 2


class VtError(RuntimeError):
    def __init__(self, code, text):
        super().__init__(f"vittrack_hip error {code}: {text}")
        self.code = code


class CBBox(Structure):
    _fields_ = [("x", c_int32), ("y", c_int32), ("width", c_int32), ("height", c_int32)]


class CResult(Structure):
    _fields_ = [("success", c_int32), ("score", c_float), ("bbox", CBBox)]


class CConfig(Structure):
    _fields_ = [("struct_size", c_uint32), ("success_threshold", c_float), ("use_graph", c_int32),
                ("n_streams", c_int32), ("max_frame_width", c_int32),
                ("max_frame_height", c_int32), ("max_device_mib", c_int32),
                ("host_window_margin_pct", c_int32), ("host_zero_copy", c_int32), ("reserved", c_int32 * 5)]


class CModelInfo(Structure):
    _fields_ = [("patch", c_int32), ("template_size", c_int32), ("search_size", c_int32),
                ("dim", c_int32), ("heads", c_int32), ("layers", c_int32), ("mlp_dim", c_int32),
                ("head_channels", c_int32), ("tokens_template", c_int32),
                ("tokens_search", c_int32), ("kpad", c_int32), ("score_grid", c_int32),
                ("flops_per_frame", c_double), ("encoder_flops_per_frame", c_double),
                ("weight_bytes", c_uint64)]


class CFrame(Structure):
    _fields_ = [("plane0", c_void_p), ("plane1", c_void_p), ("width", c_int32),
                ("height", c_int32), ("stride0", c_int32), ("stride1", c_int32),
                ("format", c_int32), ("origin_x", c_int32), ("origin_y", c_int32),
                ("windowed", c_int32), ("window_w", c_int32), ("window_h", c_int32)]


class CDrawCmd(Structure):
    _fields_ = [("type", c_int32), ("x", c_int32), ("y", c_int32), ("w", c_int32), ("h", c_int32),
                ("p", c_int32), ("value", c_int32), ("text", c_char * 36)]


DRAW_BACKGROUND, DRAW_TEXT, DRAW_RECT, DRAW_CROSSHAIR, DRAW_CURSOR, DRAW_SELECTION = range(6)


class CKernelTime(Structure):
    _fields_ = [("name", c_char * 48), ("launches", c_int32), ("ms_total", c_float),
                ("flops", c_double), ("bytes", c_double)]


_lib = None

# every symbol include/vittrack_hip.h declares (tests check the library exports all of them)
EXPORTS = [
    "vt_config_default", "vt_last_error", "vt_abi_version", "vt_build_info", "vt_device_count", "vt_create",
    "vt_create_from_device_blob", "vt_destroy", "vt_get_model_info", "vt_init_rgb8",
    "vt_update_rgb8", "vt_init_yuy2", "vt_update_yuy2", "vt_init_nv12", "vt_update_nv12", "vt_init_rgb8_device",
    "vt_update_rgb8_device", "vt_init_nv12_device", "vt_update_nv12_device", "vt_group_create",
    "vt_group_create_from_device_blob", "vt_group_destroy", "vt_group_streams",
    "vt_group_get_model_info", "vt_group_init_device", "vt_group_enqueue_device", "vt_group_wait",
    "vt_recommended_streams", "vt_plan_engines", "vt_import_dmabuf", "vt_release_dmabuf", "vt_export_dmabuf", "vt_host_register", "vt_host_unregister", "vt_group_update_device", "vt_group_hip_stream", "vt_group_init_host", "vt_group_update_host", "vt_group_enqueue_host", "vt_group_wait_next",
    "vt_group_host_redos", "vt_group_graph_captures", "vt_nv12_to_rgb8", "vt_nv12_to_rgb8_device", "vt_nv12_to_rgb8_batch_device", "vt_overlay_nv12", "vt_overlay_nv12_device", "vt_overlay_rgb8",
    "vt_overlay_rgb8_device",
    "vt_group_profile_device", "vt_group_enable_taps", "vt_group_set_tuning", "vt_group_set_state_box", "vt_tracker_as_group",
    "vt_group_read_tensor",
    "vt_rccl_unique_id", "vt_broadcast_weights_rccl", "vt_free_device_blob",
]
# every symbol include/vittrack_hip_ops.h declares (libvittrack_hip_ops.so; the product library exports none of them)
OPS_EXPORTS = [
    "vt_op_gemm_bf16", "vt_op_gemm_bench", "vt_op_qkv_bf16", "vt_op_attention_bf16",
    "vt_op_attention_bench", "vt_op_layernorm", "vt_op_nv12_to_rgb8_bench", "vt_op_nv12_to_rgb8_batch_bench", "vt_op_conv3x3_relu_bf16", "vt_op_headconv_bf16",
    "vt_op_headconv_ln_bf16",
]


def lib():
    """Load libvittrack_hip.so (built in-tree by build.py). Fails loudly if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VtError(-2, f"{LIB_PATH} not built: run `python __graft_entry__.py` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # One HIP runtime per process: PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64 and a
    # second copy loaded later cannot see the GPU ("No HIP GPUs are available"). Importing torch
    # first makes this library's DT_NEEDED libamdhip64.so.7 resolve to the copy already loaded.
    try:
        import torch  # noqa: F401
    except Exception:  # torch absent: the library brings in /opt/rocm's runtime itself
        pass
    L = ctypes.CDLL(LIB_PATH)
    L.vt_last_error.restype = c_char_p
    L.vt_build_info.restype = c_char_p
    L.vt_config_default.argtypes = [POINTER(CConfig)]
    L.vt_create.argtypes = [c_char_p, c_int, POINTER(CConfig), POINTER(c_void_p)]
    L.vt_create_from_device_blob.argtypes = [c_void_p, c_size_t, c_int, POINTER(CConfig),
                                             POINTER(c_void_p)]
    L.vt_destroy.argtypes = [c_void_p]
    L.vt_destroy.restype = None
    L.vt_get_model_info.argtypes = [c_void_p, POINTER(CModelInfo)]
    u8p = POINTER(c_uint8)
    L.vt_init_rgb8.argtypes = [c_void_p, u8p, c_int, c_int, c_int, CBBox]
    L.vt_update_rgb8.argtypes = [c_void_p, u8p, c_int, c_int, c_int, POINTER(CResult)]
    L.vt_init_yuy2.argtypes = [c_void_p, u8p, c_int, c_int, c_int, CBBox]
    L.vt_update_yuy2.argtypes = [c_void_p, u8p, c_int, c_int, c_int, POINTER(CResult)]
    L.vt_init_nv12.argtypes = [c_void_p, u8p, u8p, c_int, c_int, c_int, c_int, CBBox]
    L.vt_update_nv12.argtypes = [c_void_p, u8p, u8p, c_int, c_int, c_int, c_int, POINTER(CResult)]
    L.vt_init_rgb8_device.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, CBBox]
    L.vt_update_rgb8_device.argtypes = [c_void_p, c_void_p, c_int, c_int, c_int, POINTER(CResult)]
    L.vt_init_nv12_device.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                      CBBox]
    L.vt_update_nv12_device.argtypes = [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                        POINTER(CResult)]
    L.vt_host_register.argtypes = [c_int, c_void_p, c_size_t, POINTER(c_void_p)]
    L.vt_host_unregister.argtypes = [c_int, c_void_p]
    L.vt_group_create.argtypes = [c_char_p, c_int, POINTER(CConfig), POINTER(c_void_p)]
    L.vt_group_create_from_device_blob.argtypes = [c_void_p, c_size_t, c_int, POINTER(CConfig),
                                                   POINTER(c_void_p)]
    L.vt_group_destroy.argtypes = [c_void_p]
    L.vt_group_destroy.restype = None
    L.vt_group_streams.argtypes = [c_void_p]
    L.vt_group_get_model_info.argtypes = [c_void_p, POINTER(CModelInfo)]
    L.vt_group_init_device.argtypes = [c_void_p, c_int, POINTER(CFrame), CBBox]
    L.vt_group_enqueue_device.argtypes = [c_void_p, POINTER(CFrame), c_int]
    L.vt_group_wait.argtypes = [c_void_p, POINTER(CResult), c_int]
    L.vt_group_update_device.argtypes = [c_void_p, POINTER(CFrame), c_int, POINTER(CResult)]
    L.vt_recommended_streams.argtypes = [POINTER(CModelInfo), c_int]
    L.vt_plan_engines.argtypes = [POINTER(CModelInfo), c_int, POINTER(c_int), c_int]
    L.vt_import_dmabuf.argtypes = [c_int, c_int, c_size_t, POINTER(c_void_p), POINTER(c_void_p)]
    L.vt_release_dmabuf.argtypes = [c_void_p]
    L.vt_release_dmabuf.restype = None
    L.vt_export_dmabuf.argtypes = [c_int, c_void_p, c_size_t, POINTER(c_int)]
    L.vt_group_init_host.argtypes = [c_void_p, c_int, POINTER(CFrame), CBBox]
    L.vt_group_update_host.argtypes = [c_void_p, POINTER(CFrame), c_int, POINTER(CResult)]
    L.vt_group_enqueue_host.argtypes = [c_void_p, POINTER(CFrame), c_int]
    L.vt_group_wait_next.argtypes = [c_void_p, POINTER(CResult), c_int]
    L.vt_group_host_redos.argtypes = [c_void_p]
    L.vt_group_graph_captures.argtypes = [c_void_p]
    L.vt_group_hip_stream.argtypes = [c_void_p]
    L.vt_group_hip_stream.restype = c_void_p
    L.vt_group_profile_device.argtypes = [c_void_p, POINTER(CFrame), c_int, c_int,
                                          POINTER(CKernelTime), c_int]
    L.vt_group_enable_taps.argtypes = [c_void_p, c_int]
    L.vt_group_set_tuning.argtypes = [c_void_p, c_char_p, c_int]
    L.vt_group_set_state_box.argtypes = [c_void_p, c_int, POINTER(c_float)]
    L.vt_tracker_as_group.argtypes = [c_void_p]
    L.vt_tracker_as_group.restype = c_void_p
    L.vt_group_read_tensor.argtypes = [c_void_p, c_int, c_char_p, POINTER(c_float), c_int64]
    L.vt_group_read_tensor.restype = c_int64
    L.vt_nv12_to_rgb8.argtypes = [c_int, u8p, c_size_t, c_int, c_int, u8p]
    L.vt_nv12_to_rgb8_device.argtypes = [c_int, c_void_p, c_size_t, c_int, c_int, c_void_p,
                                         c_void_p]
    L.vt_nv12_to_rgb8_batch_device.argtypes = [c_int, POINTER(c_void_p), POINTER(c_size_t), c_int, c_int, c_int,
                                               POINTER(c_void_p), c_void_p]
    L.vt_overlay_nv12_device.argtypes = [c_int, c_void_p, c_int, c_int, c_int, POINTER(CDrawCmd), c_int,
                                         c_void_p]
    L.vt_overlay_nv12.argtypes = [c_int, u8p, c_int, c_int, POINTER(CDrawCmd), c_int]
    L.vt_overlay_rgb8_device.argtypes = L.vt_overlay_nv12_device.argtypes
    L.vt_overlay_rgb8.argtypes = L.vt_overlay_nv12.argtypes
    L.vt_rccl_unique_id.argtypes = [u8p]
    L.vt_broadcast_weights_rccl.argtypes = [u8p, c_int, c_int, c_int, c_char_p, POINTER(c_void_p),
                                            POINTER(c_size_t)]
    L.vt_free_device_blob.argtypes = [c_int, c_void_p]
    L.vt_free_device_blob.restype = None
    _lib = L
    return L


_ops = None


def ops_lib():
    """Load libvittrack_hip_ops.so: the product's objects + the operator-level entry points (tests, tuning tools)."""
    global _ops
    if _ops is not None:
        return _ops
    if not os.path.exists(OPS_LIB_PATH):
        raise VtError(-2, f"{OPS_LIB_PATH} not built: run `python __graft_entry__.py`")
    lib()       # torch / HIP runtime first, as above
    L = ctypes.CDLL(OPS_LIB_PATH)
    L.vt_last_error.restype = c_char_p
    u16p, fp = POINTER(c_uint16), POINTER(c_float)
    L.vt_op_gemm_bf16.argtypes = [c_int, u16p, u16p, fp, fp, c_int, c_int, c_int, c_int, c_int, fp, fp, fp, c_float]
    L.vt_op_gemm_bench.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, c_int, fp]
    L.vt_op_qkv_bf16.argtypes = [c_int, u16p, u16p, fp, fp, fp, c_int, c_int, c_int, c_int, c_int, fp, fp]
    L.vt_op_attention_bf16.argtypes = [c_int, u16p, u16p, u16p, fp, c_int, c_int, c_int, c_int]
    L.vt_op_attention_bench.argtypes = [c_int, c_int, c_int, c_int, c_int, c_int, fp]
    L.vt_op_layernorm.argtypes = [c_int, fp, fp, fp, fp, c_int, c_int]
    L.vt_op_conv3x3_relu_bf16.argtypes = [c_int, u16p, u16p, fp, fp, c_int, c_int, c_int, c_int, c_int]
    L.vt_op_nv12_to_rgb8_bench.argtypes = [c_int, c_int, c_int, c_int, fp]
    L.vt_op_nv12_to_rgb8_batch_bench.argtypes = [c_int, c_int, c_int, c_int, c_int, fp]
    _ops = L
    return L


def _check(rc):
    if rc < 0:
        raise VtError(rc, lib().vt_last_error().decode(errors="replace"))
    return rc


def _check_op(rc):
    if rc < 0:
        raise VtError(rc, ops_lib().vt_last_error().decode(errors="replace"))
    return rc


def build_info() -> dict:
    """vt_build_info(): 'key=value;...' of the loaded library - ABI version and the sha256 build.py stamped into the
    translation unit of the dominant kernels (their sources + compile flags)"""
    txt = lib().vt_build_info().decode()
    return dict(kv.split("=", 1) for kv in txt.split(";") if "=" in kv)


def recommended_streams(info: "CModelInfo", max_streams: int = 128) -> int:
    """vt_recommended_streams: streams per group that fill the 256 CUs in whole GEMM rounds"""
    return lib().vt_recommended_streams(byref(info), max_streams)


def model_info_for(cfg_name: str) -> "CModelInfo":
    """the fields of vt_model_info the planners read, from a named configuration (no engine needed)"""
    cfg = weights.get_config(cfg_name)
    mi = CModelInfo()
    mi.patch, mi.template_size, mi.search_size = cfg.patch, cfg.template, cfg.search
    mi.dim, mi.heads, mi.layers, mi.mlp_dim, mi.head_channels = cfg.dim, cfg.heads, cfg.layers, cfg.mlp_dim, cfg.head_ch
    mi.tokens_template, mi.tokens_search, mi.kpad, mi.score_grid = cfg.n_t, cfg.n_s, cfg.kpad, cfg.grid_s
    return mi


def plan_engines(info: "CModelInfo", n_streams: int) -> list:
    """vt_plan_engines: engine (Group) sizes for n_streams on one GPU that avoid a nearly empty GEMM round"""
    sizes = (c_int * 16)()
    k = lib().vt_plan_engines(byref(info), n_streams, sizes, 16)
    if k <= 0:
        raise ValueError(f"vt_plan_engines({n_streams}) failed")
    return [int(sizes[i]) for i in range(k)]


class DmaBuf:
    """a dma-buf mapped into device memory (vt_import_dmabuf); .ptr is usable as a frame plane"""

    def __init__(self, fd: int, nbytes: int, device: int = 0):
        self._h, p = c_void_p(), c_void_p()
        _check(lib().vt_import_dmabuf(device, fd, nbytes, byref(self._h), byref(p)))
        self.ptr, self.nbytes = p.value, nbytes

    def close(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                lib().vt_release_dmabuf(self._h)
                self._h = c_void_p()
        except TypeError:      # interpreter shutdown: module globals are already None
            pass

    __del__ = close


def export_dmabuf(d_ptr: int, nbytes: int, device: int = 0) -> int:
    fd = c_int(-1)
    _check(lib().vt_export_dmabuf(device, d_ptr, nbytes, byref(fd)))
    return fd.value


class HostMapping:
    """A host buffer (NumPy array) page-locked and mapped into the device's address space
    (vt_host_register): `.d_ptr` + offset serves as plane pointers of frame_nv12 / frame_rgb8 with the
    *_device entry points - the pixel kernel reads only the pixels it samples over PCIe."""

    def __init__(self, arr: np.ndarray, device: int = 0):
        self.arr, self.device, self.d_ptr = arr, device, None
        dp = c_void_p()
        _check(lib().vt_host_register(device, c_void_p(arr.ctypes.data), arr.nbytes, byref(dp)))
        self.d_ptr = dp.value

    def close(self):
        if self.d_ptr is not None:
            lib().vt_host_unregister(self.device, c_void_p(self.arr.ctypes.data))
            self.d_ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def rccl_unique_id() -> bytes:
    """rank 0: the 128-byte ncclUniqueId the host ships to the other ranks"""
    buf = (c_uint8 * 128)()
    _check(lib().vt_rccl_unique_id(buf))
    return bytes(buf)


def broadcast_weights_rccl(unique_id: bytes, world: int, rank: int, device: int,
                           weights_path: str | None):
    """vt_broadcast_weights_rccl: -> (device pointer, nbytes); release with free_device_blob"""
    idb = (c_uint8 * 128)(*unique_id)
    p, n = c_void_p(), c_size_t()
    _check(lib().vt_broadcast_weights_rccl(idb, world, rank, device,
                                           weights_path.encode() if weights_path else None,
                                           byref(p), byref(n)))
    return p.value, n.value


def free_device_blob(ptr: int, device: int = 0):
    lib().vt_free_device_blob(device, ptr)


def device_count() -> int:
    return lib().vt_device_count()


def _u8(a):
    return a.ctypes.data_as(POINTER(c_uint8))


def _u16(a):
    return a.ctypes.data_as(POINTER(c_uint16))


def _f32(a):
    return a.ctypes.data_as(POINTER(c_float))


def make_config(success_threshold=-1.0, use_graph=True, n_streams=1, max_w=0, max_h=0,
                max_device_mib=0, host_window_margin_pct=0, host_zero_copy=0) -> CConfig:
    c = CConfig()
    lib().vt_config_default(byref(c))
    c.success_threshold = success_threshold
    c.use_graph = 1 if use_graph else 0
    c.n_streams = n_streams
    c.max_frame_width, c.max_frame_height = max_w, max_h
    c.max_device_mib = max_device_mib
    c.host_window_margin_pct = host_window_margin_pct
    c.host_zero_copy = host_zero_copy
    return c


# ---- reference-shaped API ---------------------------------------------------------------------

class BBox:
    """≙ vit_tracker::BBox (src/selection_state.rs:44, src/tracker_context.rs:94)."""
    __slots__ = ("x", "y", "width", "height")

    def __init__(self, x, y, width, height):
        self.x, self.y, self.width, self.height = int(x), int(y), int(width), int(height)

    @staticmethod
    def new(x, y, width, height):
        return BBox(x, y, width, height)

    @staticmethod
    def from_array(a):
        return BBox(int(a[0]), int(a[1]), int(a[2]), int(a[3]))

    def as_tuple(self):
        return (self.x, self.y, self.width, self.height)

    def _c(self):
        return CBBox(self.x, self.y, self.width, self.height)

    def __eq__(self, o):
        return isinstance(o, BBox) and self.as_tuple() == o.as_tuple()

    def __repr__(self):
        return f"BBox(x={self.x}, y={self.y}, width={self.width}, height={self.height})"


class TrackResult:
    """≙ the Ok value of VitTrack::update: .success, .score, .bbox ([x, y, w, h])"""
    __slots__ = ("success", "score", "bbox")

    def __init__(self, c: CResult):
        self.success = bool(c.success)
        self.score = float(c.score)
        self.bbox = [c.bbox.x, c.bbox.y, c.bbox.width, c.bbox.height]

    def __repr__(self):
        return f"TrackResult(success={self.success}, score={self.score:.4f}, bbox={self.bbox})"


class NV12Frame:
    """packed NV12 host buffer: Y plane then interleaved UV, stride == width
    (the layout /root/reference/src/nv12_convert.rs:47-54 assumes)"""

    def __init__(self, buf: np.ndarray, width: int, height: int):
        self.buf = np.ascontiguousarray(buf, np.uint8).reshape(-1)
        self.w, self.h = width, height
        assert self.buf.size >= width * height + ((width + 1) & ~1) * ((height + 1) // 2)


class YUY2Frame:
    """packed 4:2:2 host frame (Y0 U Y1 V), rows of 2*width bytes (src/pipeline_ir.rs:27-41)"""

    def __init__(self, buf: np.ndarray, width: int, height: int):
        self.buf = np.ascontiguousarray(buf, np.uint8).reshape(-1)
        self.w, self.h = width, height
        assert self.buf.size >= 2 * width * height and width % 2 == 0


class VitTrack:
    """≙ vit_tracker::VitTrack (src/tracker_context.rs:21,88,90,120)."""

    def __init__(self, weights_path: str, device: int = 0, success_threshold: float = -1.0,
                 use_graph: bool = True, max_w: int = 0, max_h: int = 0):
        self._h = c_void_p()
        cfg = make_config(success_threshold, use_graph, 1, max_w, max_h)
        _check(lib().vt_create(weights_path.encode(), device, byref(cfg), byref(self._h)))

    @staticmethod
    def new(model_path: str, **kw) -> "VitTrack":
        return VitTrack(model_path, **kw)

    def close(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                lib().vt_destroy(self._h)
                self._h = c_void_p()
        except TypeError:      # interpreter shutdown: module globals are already None
            pass

    __del__ = close

    def model_info(self) -> CModelInfo:
        mi = CModelInfo()
        _check(lib().vt_get_model_info(self._h, byref(mi)))
        return mi

    def as_group(self) -> "Group":
        return Group._view(lib().vt_tracker_as_group(self._h), self)

    def init(self, frame, bbox: BBox) -> None:
        """frame: (H,W,3) uint8 RGB array (≙ ArrayView3<u8>) or NV12Frame. Like the reference's
        call site (src/tracker_context.rs:88) the caller gets nothing back; errors raise."""
        if isinstance(frame, YUY2Frame):
            _check(lib().vt_init_yuy2(self._h, _u8(frame.buf), frame.w, frame.h, 2 * frame.w,
                                      bbox._c()))
        elif isinstance(frame, NV12Frame):
            y = frame.buf
            uv = frame.buf[frame.w * frame.h:]
            _check(lib().vt_init_nv12(self._h, _u8(y), _u8(uv), frame.w, frame.h, frame.w,
                                      (frame.w + 1) & ~1, bbox._c()))
        else:
            a = np.ascontiguousarray(frame, np.uint8)
            h, w, _ = a.shape
            _check(lib().vt_init_rgb8(self._h, _u8(a), w, h, w * 3, bbox._c()))

    def update(self, frame) -> TrackResult:
        r = CResult()
        if isinstance(frame, YUY2Frame):
            _check(lib().vt_update_yuy2(self._h, _u8(frame.buf), frame.w, frame.h, 2 * frame.w,
                                        byref(r)))
        elif isinstance(frame, NV12Frame):
            y = frame.buf
            uv = frame.buf[frame.w * frame.h:]
            _check(lib().vt_update_nv12(self._h, _u8(y), _u8(uv), frame.w, frame.h, frame.w,
                                        (frame.w + 1) & ~1, byref(r)))
        else:
            a = np.ascontiguousarray(frame, np.uint8)
            h, w, _ = a.shape
            _check(lib().vt_update_rgb8(self._h, _u8(a), w, h, w * 3, byref(r)))
        return TrackResult(r)

    # device-resident frames (pointers into this GPU's HBM, e.g. torch tensors' data_ptr())
    def init_nv12_device(self, d_y, d_uv, w, h, y_stride, uv_stride, bbox: BBox):
        _check(lib().vt_init_nv12_device(self._h, d_y, d_uv, w, h, y_stride, uv_stride, bbox._c()))

    def update_nv12_device(self, d_y, d_uv, w, h, y_stride, uv_stride) -> TrackResult:
        r = CResult()
        _check(lib().vt_update_nv12_device(self._h, d_y, d_uv, w, h, y_stride, uv_stride,
                                           byref(r)))
        return TrackResult(r)

    def init_rgb8_device(self, d_rgb, w, h, stride, bbox: BBox):
        _check(lib().vt_init_rgb8_device(self._h, d_rgb, w, h, stride, bbox._c()))

    def update_rgb8_device(self, d_rgb, w, h, stride) -> TrackResult:
        r = CResult()
        _check(lib().vt_update_rgb8_device(self._h, d_rgb, w, h, stride, byref(r)))
        return TrackResult(r)


def frame_nv12(d_y, d_uv, w, h, y_stride=None, uv_stride=None) -> CFrame:
    return CFrame(d_y, d_uv, w, h, y_stride or w, uv_stride or ((w + 1) & ~1), PIX_NV12, 0, 0, 0, 0, 0)


def frame_rgb8(d_rgb, w, h, stride=None) -> CFrame:
    return CFrame(d_rgb, None, w, h, stride or 3 * w, 0, PIX_RGB8, 0, 0, 0, 0, 0)


class Group:
    """B independent tracked streams batched on one GPU (vt_group_*)."""

    def __init__(self, weights_path: str | None = None, n_streams: int = 1, device: int = 0,
                 success_threshold: float = -1.0, use_graph: bool = True,
                 device_blob: tuple[int, int] | None = None, max_device_mib: int = 0,
                 host_window_margin_pct: int = 0, host_zero_copy: int = 0):
        self._h = c_void_p()
        self._owner = None
        self._keep = {}
        cfg = make_config(success_threshold, use_graph, n_streams, max_device_mib=max_device_mib,
                          host_window_margin_pct=host_window_margin_pct, host_zero_copy=host_zero_copy)
        if device_blob is not None:
            ptr, nbytes = device_blob
            _check(lib().vt_group_create_from_device_blob(ptr, nbytes, device, byref(cfg),
                                                          byref(self._h)))
        else:
            _check(lib().vt_group_create(weights_path.encode(), device, byref(cfg),
                                         byref(self._h)))
        self._own = True

    @classmethod
    def _view(cls, handle, owner):
        g = cls.__new__(cls)
        g._h = c_void_p(handle)
        g._owner = owner
        g._own = False
        return g

    def close(self):
        try:
            if getattr(self, "_own", False) and self._h.value:
                lib().vt_group_destroy(self._h)
            self._h = c_void_p()
        except TypeError:      # interpreter shutdown: module globals are already None
            pass

    __del__ = close

    @property
    def streams(self) -> int:
        return lib().vt_group_streams(self._h)

    def model_info(self) -> CModelInfo:
        mi = CModelInfo()
        _check(lib().vt_group_get_model_info(self._h, byref(mi)))
        return mi

    def hip_stream(self) -> int:
        return lib().vt_group_hip_stream(self._h)

    def init_device(self, stream: int, frame: CFrame, bbox: BBox):
        _check(lib().vt_group_init_device(self._h, stream, byref(frame), bbox._c()))

    @staticmethod
    def _arr(frames):
        arr = (CFrame * len(frames))(*frames)
        return arr

    def enqueue_device(self, frames):
        arr = self._arr(frames)
        _check(lib().vt_group_enqueue_device(self._h, arr, len(frames)))

    def wait(self):
        n = self.streams
        out = (CResult * n)()
        _check(lib().vt_group_wait(self._h, out, n))
        return [TrackResult(r) for r in out]

    def update_device(self, frames):
        arr = self._arr(frames)
        out = (CResult * len(frames))()
        _check(lib().vt_group_update_device(self._h, arr, len(frames), out))
        return [TrackResult(r) for r in out]

    @staticmethod
    def _host_frame(frame):
        """(CFrame with HOST pointers, keep-alive object) for an (H,W,3) RGB array, NV12Frame or
        YUY2Frame"""
        if isinstance(frame, NV12Frame):
            uv = frame.buf[frame.w * frame.h:]
            return CFrame(frame.buf.ctypes.data, uv.ctypes.data, frame.w, frame.h, frame.w,
                          (frame.w + 1) & ~1, PIX_NV12, 0, 0, 0, 0, 0), frame
        if isinstance(frame, YUY2Frame):
            return CFrame(frame.buf.ctypes.data, None, frame.w, frame.h, 2 * frame.w, 0, PIX_YUY2,
                          0, 0, 0, 0, 0), frame
        a = np.ascontiguousarray(frame, np.uint8)
        h, w, _ = a.shape
        return CFrame(a.ctypes.data, None, w, h, 3 * w, 0, PIX_RGB8, 0, 0, 0, 0, 0), a

    def init_host(self, stream: int, frame, bbox: BBox):
        f, keep = self._host_frame(frame)
        _check(lib().vt_group_init_host(self._h, stream, byref(f), bbox._c()))

    def update_host(self, frames):
        """one pass on HOST frames (RGB arrays / NV12Frame / YUY2Frame, one per stream): only the
        search windows cross PCIe, in one copy"""
        pairs = [self._host_frame(fr) for fr in frames]
        arr = (CFrame * len(pairs))(*[p[0] for p in pairs])
        out = (CResult * len(pairs))()
        _check(lib().vt_group_update_host(self._h, arr, len(pairs), out))
        return [TrackResult(r) for r in out]

    def enqueue_host(self, frames):
        """pipelined host pass: returns once the windows are packed and the upload + pass are
        enqueued; collect with wait_next(). The frames must stay alive and unchanged until then
        (this wrapper keeps references)."""
        pairs = [self._host_frame(fr) for fr in frames]
        arr = (CFrame * len(pairs))(*[p[0] for p in pairs])
        _check(lib().vt_group_enqueue_host(self._h, arr, len(pairs)))
        if not hasattr(self, "_keep") or self._keep is None:
            self._keep = {}
        self._keep[self._keep.get("seq", 0)] = pairs
        self._keep["seq"] = self._keep.get("seq", 0) + 1

    def wait_next(self):
        n = self.streams
        out = (CResult * n)()
        _check(lib().vt_group_wait_next(self._h, out, n))
        if getattr(self, "_keep", None):
            done = self._keep.get("done", 0)
            self._keep.pop(done, None)
            self._keep["done"] = done + 1
        return [TrackResult(r) for r in out]

    def graph_captures(self) -> int:
        """hipGraph captures since creation: all crop tiers are captured when the engine is created, none inside a pass"""
        return lib().vt_group_graph_captures(self._h)

    def host_redos(self) -> int:
        return lib().vt_group_host_redos(self._h)

    def profile_device(self, frames, iters=5):
        arr = self._arr(frames)
        out = (CKernelTime * 64)()
        n = _check(lib().vt_group_profile_device(self._h, arr, len(frames), iters, out, 64))
        return [dict(name=out[i].name.decode(), launches=out[i].launches,
                     ms=float(out[i].ms_total), flops=out[i].flops, bytes=out[i].bytes)
                for i in range(n)]

    def set_state_box(self, stream: int, box):
        b = (c_float * 4)(*[float(v) for v in box])
        _check(lib().vt_group_set_state_box(self._h, stream, b))

    def enable_taps(self, on=True):
        _check(lib().vt_group_enable_taps(self._h, 1 if on else 0))

    def set_tuning(self, key: str, value: int):
        """diagnostics: alternative kernels for A/B runs (vt_group_set_tuning)"""
        _check(lib().vt_group_set_tuning(self._h, key.encode(), int(value)))

    def read_tensor(self, name: str, stream: int = 0) -> np.ndarray:
        n = _check(lib().vt_group_read_tensor(self._h, stream, name.encode(), None, 0))
        out = np.empty(n, np.float32)
        _check(lib().vt_group_read_tensor(self._h, stream, name.encode(), _f32(out), n))
        return out

    def read_state(self, stream: int = 0) -> dict:
        raw = self.read_tensor("state", stream)
        i = raw.view(np.int32)
        return dict(box=raw[0:4].copy(), geo=raw[4:8].copy(), frame_w=int(i[8]),
                    frame_h=int(i[9]), initialized=int(i[10]), frames_done=int(i[11]),
                    success_count=int(i[12]), last_idx=int(i[13]), last_fbox=raw[14:18].copy(),
                    last_score=float(raw[18]))


# ---- reference colour converter -------------------------------------------------------------

def nv12_full_to_rgb(nv12_data: np.ndarray, width: int, height: int, device: int = 0):
    """≙ nv12_full_to_rgb_parallel (src/nv12_convert.rs:46) on the GPU -> (H,W,3) uint8"""
    buf = np.ascontiguousarray(nv12_data, np.uint8).reshape(-1)
    out = np.empty((height, width, 3), np.uint8)
    _check(lib().vt_nv12_to_rgb8(device, _u8(buf), buf.size, width, height, _u8(out)))
    return out


# ---- overlays (the reference's per-frame drawing, on the GPU) -------------------------------------

def nv12_to_rgb8_batch_device(d_nv12_ptrs, lens, w: int, h: int, d_rgb_ptrs, device: int = 0, hip_stream=None):
    """vt_nv12_to_rgb8_batch_device: n device-resident packed NV12 frames -> n RGB8 frames in one launch per 64 frames"""
    n = len(d_nv12_ptrs)
    ins = (c_void_p * n)(*[int(p) for p in d_nv12_ptrs])
    outs = (c_void_p * n)(*[int(p) for p in d_rgb_ptrs])
    ls = (c_size_t * n)(*[int(x) for x in lens])
    _check(lib().vt_nv12_to_rgb8_batch_device(device, ins, ls, n, w, h, outs, hip_stream))


def draw_cmd(kind, x=0, y=0, w=0, h=0, p=0, value=0, text="") -> CDrawCmd:
    return CDrawCmd(kind, x, y, w, h, p, value, text.encode()[:35])


def overlay_nv12(nv12: np.ndarray, width: int, height: int, cmds, device: int = 0) -> np.ndarray:
    """apply draw commands to the luma plane of a packed NV12 host buffer (returns a copy)"""
    buf = np.ascontiguousarray(nv12, np.uint8).reshape(-1).copy()
    arr = (CDrawCmd * len(cmds))(*cmds)
    _check(lib().vt_overlay_nv12(device, _u8(buf), width, height, arr, len(cmds)))
    return buf


def overlay_rgb8(rgb: np.ndarray, cmds, device: int = 0) -> np.ndarray:
    """apply draw commands to an (H,W,3) RGB8 host image (returns a copy)"""
    img = np.ascontiguousarray(rgb, np.uint8).copy()
    h, w, _ = img.shape
    arr = (CDrawCmd * len(cmds))(*cmds)
    _check(lib().vt_overlay_rgb8(device, _u8(img), w, h, arr, len(cmds)))
    return img


# ---- operator-level entry points (numerics tests) -------------------------------------------

def op_gemm_bf16(a_bits, w_bits, bias, c_init=None, epilogue=0, device=0, cfg=-1, rowstat=None,
                 colsum=None, want_rowstat=False, eps=1e-6):
    """vt_op_gemm_bf16. epilogue 0 / 1 / 4: the X-epilogues (x comes back as the value of the 3-byte pair the
    engine stores; want_rowstat: also the finalized (rstd, -mean * rstd) per row -> (x, rowstat));
    2 / 3: GELU / ReLU to bf16, with a folded LayerNorm if rowstat [M,2] and colsum [N] are given"""
    a_bits = np.ascontiguousarray(a_bits, np.uint16)
    w_bits = np.ascontiguousarray(w_bits, np.uint16)
    M, K = a_bits.shape
    N = w_bits.shape[0]
    c = np.zeros((M, N), np.float32) if c_init is None else np.ascontiguousarray(c_init,
                                                                                 np.float32).copy()
    b = None if bias is None else np.ascontiguousarray(bias, np.float32)
    rs = None if rowstat is None else np.ascontiguousarray(rowstat, np.float32)
    cs = None if colsum is None else np.ascontiguousarray(colsum, np.float32)
    ro = np.zeros((M, 2), np.float32) if want_rowstat else None
    _check_op(ops_lib().vt_op_gemm_bf16(device, _u16(a_bits), _u16(w_bits),
                                 _f32(b) if b is not None else None, _f32(c), M, N, K, epilogue, cfg,
                                 _f32(rs) if rs is not None else None, _f32(cs) if cs is not None else None,
                                 _f32(ro) if ro is not None else None, eps))
    return (c, ro) if want_rowstat else c


def op_gemm_bench(M, N, K, epilogue, cfg=-1, iters=50, device=0) -> float:
    us = c_float()
    _check_op(ops_lib().vt_op_gemm_bench(device, M, N, K, epilogue, cfg, iters, byref(us)))
    return float(us.value)


def op_qkv_bf16(a_bits, w_bits, bias, B, tokens, D, device=0, cfg=-1, vt_perm=0, rowstat=None,
                colsum=None):
    a_bits = np.ascontiguousarray(a_bits, np.uint16)
    w_bits = np.ascontiguousarray(w_bits, np.uint16)
    bias = np.ascontiguousarray(bias, np.float32)
    npad = (tokens + 63) // 64 * 64
    qk = np.empty((B * tokens, 2 * D), np.float32)
    vt = np.empty((B * (D // 64), 64, npad), np.float32)
    rs = None if rowstat is None else np.ascontiguousarray(rowstat, np.float32)
    cs = None if colsum is None else np.ascontiguousarray(colsum, np.float32)
    _check_op(ops_lib().vt_op_qkv_bf16(device, _u16(a_bits), _u16(w_bits), _f32(bias), _f32(qk),
                                _f32(vt), B, tokens, D, cfg, vt_perm,
                                _f32(rs) if rs is not None else None, _f32(cs) if cs is not None else None))
    return qk, vt


def op_attention_bf16(q_bits, k_bits, v_bits, B, N, H, device=0, mode=-1):
    q_bits, k_bits, v_bits = (np.ascontiguousarray(x, np.uint16) for x in (q_bits, k_bits, v_bits))
    out = np.empty((B * N, H * 64), np.float32)
    _check_op(ops_lib().vt_op_attention_bf16(device, _u16(q_bits), _u16(k_bits), _u16(v_bits), _f32(out),
                                      B, N, H, mode))
    return out


def op_attention_bench(B, N, H, mode=-1, iters=30, device=0) -> float:
    us = c_float()
    _check_op(ops_lib().vt_op_attention_bench(device, B, N, H, mode, iters, byref(us)))
    return float(us.value)


def op_nv12_to_rgb8_bench(w, h, iters=50, device=0) -> float:
    """mean microseconds per launch of the whole-frame NV12 -> RGB8 converter (device-resident)"""
    us = c_float()
    _check_op(ops_lib().vt_op_nv12_to_rgb8_bench(device, w, h, iters, byref(us)))
    return float(us.value)


def op_nv12_to_rgb8_batch_bench(w, h, n, iters=20, device=0) -> float:
    """mean microseconds per launch of the n-frames-per-launch converter on device-resident random frames"""
    us = c_float(0.0)
    _check_op(ops_lib().vt_op_nv12_to_rgb8_batch_bench(device, w, h, n, iters, byref(us)))
    return float(us.value)


def op_conv3x3_relu(t_bf16_bits, w_bf16_bits, bias, B, grid, cfg=-1, device=0):
    """vt_op_conv3x3_relu_bf16: relu(conv3x3(t) + bias) as the implicit GEMM of the head; t [B*grid*grid][C],
    w [N][9*C] as uint16 bf16 bit patterns -> [B*grid*grid][N] float32 (bf16 values)"""
    t = np.ascontiguousarray(t_bf16_bits, np.uint16)
    w = np.ascontiguousarray(w_bf16_bits, np.uint16)
    C, N = t.shape[1], w.shape[0]
    out = np.empty((t.shape[0], N), np.float32)
    _check_op(ops_lib().vt_op_conv3x3_relu_bf16(device, t.ctypes.data_as(POINTER(ctypes.c_uint16)),
                                         w.ctypes.data_as(POINTER(ctypes.c_uint16)),
                                         _f32(np.ascontiguousarray(bias, np.float32)), _f32(out), B, grid, C, N, cfg))
    return out


def op_headconv(t_bf16_bits, w_bf16_bits, bias, B, grid, conv3x3=True, R=0, ncb=0, device=0):
    """vt_op_headconv_bf16: the head's band kernel (k_head.hip) on given operands -> [B*grid*grid][N] float32"""
    t = np.ascontiguousarray(t_bf16_bits, np.uint16)
    w = np.ascontiguousarray(w_bf16_bits, np.uint16)
    Cin, N = t.shape[1], w.shape[0]
    out = np.empty((t.shape[0], N), np.float32)
    u16 = POINTER(ctypes.c_uint16)
    _check_op(ops_lib().vt_op_headconv_bf16(device, t.ctypes.data_as(u16), w.ctypes.data_as(u16),
                                     _f32(np.ascontiguousarray(bias, np.float32)), _f32(out), B, grid, Cin, N,
                                     1 if conv3x3 else 0, R, ncb, 0, None))
    return out


def op_headconv_ln(xh_bits, xl_lo8, gamma, beta, w_bf16_bits, bias, B, grid, ntok, off, fused=True, eps=1e-6, R=0, ncb=0,
                   device=0):
    """vt_op_headconv_ln_bf16: relu(LayerNorm(xh + lo8 * 2^-12)[search rows] . w^T + bias) -> [B*grid*grid][N] float32 (the
    residual pair of specification v3: bf16 bits + signed bytes); fused: one launch (the band kernel normalises its rows
    itself), else the LayerNorm kernel followed by the band kernel"""
    xh = np.ascontiguousarray(xh_bits, np.uint16)
    xl = np.ascontiguousarray(xl_lo8, np.int8)
    w = np.ascontiguousarray(w_bf16_bits, np.uint16)
    D, N = xh.shape[1], w.shape[0]
    assert xh.shape == xl.shape == (B * ntok, D) and w.shape[1] == D
    out = np.empty((B * grid * grid, N), np.float32)
    u16 = POINTER(ctypes.c_uint16)
    _check_op(ops_lib().vt_op_headconv_ln_bf16(device, xh.ctypes.data_as(u16), xl.ctypes.data_as(POINTER(ctypes.c_int8)),
                                        _f32(np.ascontiguousarray(gamma, np.float32)),
                                        _f32(np.ascontiguousarray(beta, np.float32)), c_float(eps), ntok, off,
                                        w.ctypes.data_as(u16), _f32(np.ascontiguousarray(bias, np.float32)), _f32(out),
                                        B, grid, D, N, 1 if fused else 0, R, ncb, 0, None))
    return out


def op_headconv_ln_bench(B, grid, D, N, ntok, off, fused=True, R=0, ncb=0, iters=50, device=0) -> float:
    """mean microseconds of the head's first layer with the final LayerNorm (fused: one launch, else two)"""
    us = c_float()
    _check_op(ops_lib().vt_op_headconv_ln_bf16(device, None, None, None, None, c_float(1e-6), ntok, off, None, None, None, B, grid,
                                        D, N, 1 if fused else 0, R, ncb, iters, byref(us)))
    return float(us.value)


def op_headconv_bench(B, grid, Cin, N, conv3x3=True, R=0, ncb=0, iters=50, device=0) -> float:
    """mean microseconds per launch of the band kernel on pseudo-random operands"""
    us = c_float()
    _check_op(ops_lib().vt_op_headconv_bf16(device, None, None, None, None, B, grid, Cin, N, 1 if conv3x3 else 0, R, ncb,
                                     iters, byref(us)))
    return float(us.value)


def op_layernorm(x, gamma, beta, device=0):
    x = np.ascontiguousarray(x, np.float32)
    g = np.ascontiguousarray(gamma, np.float32)
    b = np.ascontiguousarray(beta, np.float32)
    y = np.empty_like(x)
    _check_op(ops_lib().vt_op_layernorm(device, _f32(x), _f32(g), _f32(b), _f32(y), x.shape[0],
                                 x.shape[1]))
    return y
