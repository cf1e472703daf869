"""ctypes bindings of libvittrack_host.so (include/vittrack_host.h): the C++ mirror of the
reference's TrackerContext / SelectionState / TimingStats, driven from Python tests and tools."""
from __future__ import annotations

import ctypes
import os
from ctypes import (CFUNCTYPE, POINTER, Structure, byref, c_char_p, c_double, c_float, c_int,
                    c_int32, c_size_t, c_uint8, c_uint64, c_void_p)

import numpy as np

from gstreamer_vit_tracker_amd import CBBox, CResult, VtError

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG_DIR, "libvittrack_host.so")

MOVE_UP, MOVE_DOWN, MOVE_LEFT, MOVE_RIGHT, CONFIRM, CANCEL, QUIT = range(7)
SELECTING, TRACKING, LOST = 0, 1, 2

INIT_CB = CFUNCTYPE(None, c_void_p, POINTER(c_uint8), c_int, c_int, c_int, c_int, CBBox)
UPDATE_CB = CFUNCTYPE(c_int, c_void_p, POINTER(c_uint8), c_int, c_int, c_int, c_int,
                      POINTER(CResult))


class CCallbacks(Structure):
    _fields_ = [("init", INIT_CB), ("update", UPDATE_CB)]


class CCtxInfo(Structure):
    _fields_ = [("state_kind", c_int32), ("lost_frames", c_uint64), ("has_bbox", c_int32),
                ("current_bbox", CBBox), ("current_score", c_float), ("pending_confirm", c_int32),
                ("cursor_x", c_int32), ("cursor_y", c_int32), ("start_x", c_int32),
                ("start_y", c_int32), ("selection_phase", c_int32), ("frame_width", c_int32),
                ("frame_height", c_int32)]


EXPORTS = ["vth_last_error", "vth_ctx_new", "vth_ctx_new_with_tracker", "vth_ctx_free",
           "vth_ctx_handle_command", "vth_ctx_process_frame_rgb8", "vth_ctx_process_frame_nv12",
           "vth_ctx_state_name", "vth_ctx_get_info", "vth_selection_bbox", "vth_timing_new",
           "vth_timing_free", "vth_timing_add_interval", "vth_timing_add_times", "vth_timing_fps",
           "vth_timing_avg_conv_ms", "vth_timing_avg_track_ms", "vth_nv12_full_to_rgb"]

_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VtError(-2, f"{LIB_PATH} not built: run `python __graft_entry__.py`")
    try:
        import torch  # noqa: F401  (one HIP runtime per process, see __init__.lib)
    except Exception:
        pass
    L = ctypes.CDLL(LIB_PATH)
    u8p = POINTER(c_uint8)
    L.vth_last_error.restype = c_char_p
    L.vth_ctx_new.argtypes = [c_char_p, c_int, c_int, c_int, POINTER(c_void_p)]
    L.vth_ctx_new_with_tracker.argtypes = [CCallbacks, c_void_p, c_int, c_int, POINTER(c_void_p)]
    L.vth_ctx_free.argtypes = [c_void_p]
    L.vth_ctx_free.restype = None
    L.vth_ctx_handle_command.argtypes = [c_void_p, c_int, c_int]
    L.vth_ctx_handle_command.restype = None
    L.vth_ctx_process_frame_rgb8.argtypes = [c_void_p, u8p, c_int, c_int, c_int, POINTER(CBBox)]
    L.vth_ctx_process_frame_nv12.argtypes = [c_void_p, u8p, c_int, c_int, POINTER(CBBox)]
    L.vth_ctx_state_name.argtypes = [c_void_p]
    L.vth_ctx_state_name.restype = c_char_p
    L.vth_ctx_get_info.argtypes = [c_void_p, POINTER(CCtxInfo)]
    L.vth_ctx_get_info.restype = None
    L.vth_selection_bbox.argtypes = [c_int] * 4
    L.vth_selection_bbox.restype = CBBox
    L.vth_timing_new.restype = c_void_p
    L.vth_timing_free.argtypes = [c_void_p]
    L.vth_timing_free.restype = None
    L.vth_timing_add_interval.argtypes = [c_void_p, c_uint64]
    L.vth_timing_add_interval.restype = None
    L.vth_timing_add_times.argtypes = [c_void_p, c_uint64, c_uint64]
    L.vth_timing_add_times.restype = None
    for f in ("vth_timing_fps", "vth_timing_avg_conv_ms", "vth_timing_avg_track_ms"):
        getattr(L, f).argtypes = [c_void_p]
        getattr(L, f).restype = c_double
    L.vth_nv12_full_to_rgb.argtypes = [c_int, u8p, c_size_t, c_int, c_int, u8p]
    _lib = L
    return L


class TrackerContext:
    """≙ TrackerContext (src/tracker_context.rs:7-167) — the C++ mirror behind its C ABI."""

    def __init__(self, model_path=None, width=1920, height=1080, device=0, tracker=None):
        self._h = c_void_p()
        self._keep = None
        if tracker is None:
            rc = lib().vth_ctx_new(model_path.encode(), width, height, device, byref(self._h))
        else:
            # tracker: object with init(frame_info, bbox) and update(frame_info) ->
            # (success, score, (x,y,w,h)) or raising (the reference's Err arm)
            def _init(_u, data, w, h, stride, fmt, box):
                tracker.init((data, w, h, stride, fmt), (box.x, box.y, box.width, box.height))

            def _update(_u, data, w, h, stride, fmt, out):
                try:
                    ok, score, bb = tracker.update((data, w, h, stride, fmt))
                except Exception:
                    return 1
                out.contents.success = 1 if ok else 0
                out.contents.score = score
                out.contents.bbox = CBBox(*[int(v) for v in bb])
                return 0

            self._keep = CCallbacks(INIT_CB(_init), UPDATE_CB(_update))
            rc = lib().vth_ctx_new_with_tracker(self._keep, None, width, height, byref(self._h))
        if rc < 0:
            raise VtError(rc, lib().vth_last_error().decode(errors="replace"))

    @staticmethod
    def new(model_path, width, height, device=0):
        return TrackerContext(model_path, width, height, device)

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().vth_ctx_free(self._h)
            self._h = c_void_p()

    __del__ = close

    def handle_command(self, cmd, fast=False):
        lib().vth_ctx_handle_command(self._h, cmd, 1 if fast else 0)

    def process_frame(self, frame):
        """frame: (H,W,3) uint8 RGB array, or ('nv12', packed_buffer, w, h). -> bbox tuple | None"""
        out = CBBox()
        if isinstance(frame, tuple) and frame[0] == "nv12":
            buf = np.ascontiguousarray(frame[1], np.uint8)
            rc = lib().vth_ctx_process_frame_nv12(self._h, buf.ctypes.data_as(POINTER(c_uint8)),
                                                  frame[2], frame[3], byref(out))
        else:
            a = np.ascontiguousarray(frame, np.uint8)
            h, w, _ = a.shape
            rc = lib().vth_ctx_process_frame_rgb8(self._h, a.ctypes.data_as(POINTER(c_uint8)), w,
                                                  h, 3 * w, byref(out))
        if rc < 0:
            raise VtError(rc, lib().vth_last_error().decode(errors="replace"))
        return (out.x, out.y, out.width, out.height) if rc == 1 else None

    def state_name(self):
        return lib().vth_ctx_state_name(self._h).decode()

    def info(self) -> CCtxInfo:
        i = CCtxInfo()
        lib().vth_ctx_get_info(self._h, byref(i))
        return i


class TimingStats:
    def __init__(self):
        self._h = c_void_p(lib().vth_timing_new())

    def __del__(self):
        if getattr(self, "_h", None) and self._h.value:
            lib().vth_timing_free(self._h)
            self._h = c_void_p()

    def add_interval(self, us):
        lib().vth_timing_add_interval(self._h, int(us))

    def add_times(self, conv, track):
        lib().vth_timing_add_times(self._h, int(conv), int(track))

    def fps(self):
        return lib().vth_timing_fps(self._h)

    def avg_conv_ms(self):
        return lib().vth_timing_avg_conv_ms(self._h)

    def avg_track_ms(self):
        return lib().vth_timing_avg_track_ms(self._h)


def selection_bbox(sx, sy, cx, cy):
    b = lib().vth_selection_bbox(sx, sy, cx, cy)
    return (b.x, b.y, b.width, b.height)
