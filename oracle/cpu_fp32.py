"""The CPU baseline BASELINE.md section 2 plans: the tracker's `update` as plain float32 on ALL host
cores - crop / resize / normalise by the C restatement (oracle/vt_oracle.c), the network by
torch-CPU in float32 (oracle/torch_ref.py's textbook formulation: layer_norm, linear,
scaled_dot_product_attention, erf gelu, conv2d; MKL / oneDNN threads), box decode by the C
restatement. No bf16 rounding emulation: this is what a CPU deployment of the model would run, and
it is what `bench.py`'s `cpu_baseline` times (kind "port": the reference's own tracker is a Rockchip
NPU model that cannot run on x86, /root/reference/src/main.rs:25, Cargo.toml:24).

TEST / BENCH INFRASTRUCTURE ONLY (see oracle/__init__.py): nothing in the product package imports
this file. The checker of the parity tests stays oracle/vit_ref.py (same rounding points as the HIP
kernels); tests/test_oracle_crosscheck.py holds this float32 path against it (same cell, boxes
within a pixel) so that the timed baseline is known to compute the same tracker.

Call shape: VitTrack::{init, update} as the reference host uses them
(/root/reference/src/tracker_context.rs:88,90,120).
"""
from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import vit_ref
from .torch_ref import TorchModel


class VitTrackFp32(vit_ref.VitTrackRef):
    """VitTrackRef with the network evaluated by torch-CPU in float32 (no rounding emulation)."""

    def __init__(self, weights_path, success_threshold=None, threads: int | None = None):
        super().__init__(weights_path, success_threshold)
        if threads:
            torch.set_num_threads(int(threads))
        self.threads = torch.get_num_threads()
        self.net = TorchModel(weights_path, dtype=torch.float32)

    def update(self, frame: vit_ref.Frame, taps=False) -> vit_ref.Result:
        if self.box is None:
            raise RuntimeError("update before init")
        m = self.m
        geo = vit_ref.crop_geometry(self.box, 4.0, m.S)
        srch = self._pre(frame, self.box, 4.0, m.S)
        patches = np.concatenate([self.tpl, srch], axis=0)
        out = self.net.forward_head_only(patches)
        ho = np.ascontiguousarray(out, np.float32)
        hann = np.ascontiguousarray(m.t["hann"].reshape(-1), np.float32)
        dec = np.zeros(6, np.float32)
        ib = np.zeros(4, np.int32)
        vit_ref.lib().vto_decode(vit_ref._fp(ho), vit_ref._fp(hann), m.gs, vit_ref._fp(geo), frame.w,
                                 frame.h, vit_ref._fp(dec), ib.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)))
        score = float(dec[0])
        success = bool(score >= self.thr)
        if success:
            self.box = ib.astype(np.float32).copy()
        return vit_ref.Result(success, score, tuple(int(v) for v in ib), dec[1:5].copy(), int(dec[5]))


def cpu_model_name() -> str:
    """`model name` of /proc/cpuinfo (printed beside the baseline)"""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"
