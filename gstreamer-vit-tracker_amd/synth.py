"""Deterministic synthetic moving-square clips (BASELINE.md §2 inputs; no external data).

Scene: low-amplitude seeded noise background (Y about 60 +- 16) with one bright, lightly textured
square (Y about 200) on a smooth path (<= 4 px/frame); chroma is neutral 128 except inside the
square, so the NV12 chroma path is exercised. Frames come as packed NV12 (Y plane then
interleaved UV plane, stride == width, the layout /root/reference/src/nv12_convert.rs:47-54
assumes) or as packed RGB8.
"""
from __future__ import annotations

import numpy as np


class MovingSquare:
    def __init__(self, width=1920, height=1080, square=64, seed=0, period=300, amp=None,
                 path="lissajous", center=None, hide=None):
        """hide = (t0, t1): the target is absent from frames t0 <= t < t1 (occlusion; gt_box still
        reports where it would be)"""
        self.w, self.h, self.sq, self.seed, self.period, self.path = width, height, square, seed, \
            period, path
        self.hide = tuple(hide) if hide is not None else None
        rng = np.random.default_rng(seed)
        self.bg_y = rng.integers(44, 77, size=(height, width), dtype=np.uint8)
        self.sq_y = (200 + rng.integers(-12, 13, size=(square, square))).astype(np.uint8)
        self.sq_u, self.sq_v = 90, 200
        self.cx0, self.cy0 = center if center is not None else (width / 2.0, height / 2.0)
        if amp is None:
            # max speed = amp * 2*pi / period  -> keep it under 4 px/frame
            amp = min(3.8 * period / (2 * np.pi), 0.3 * min(width, height))
        self.amp = amp
        self.phase = 0.37 * seed

    def center(self, t: int):
        a = 2 * np.pi * t / self.period + self.phase
        if self.path == "circle":
            return self.cx0 + self.amp * np.cos(a), self.cy0 + self.amp * np.sin(a)
        return (self.cx0 + self.amp * np.sin(a),
                self.cy0 + 0.6 * self.amp * np.sin(2 * a + 0.5))

    def gt_box(self, t: int):
        cx, cy = self.center(t)
        x = int(np.floor(cx - self.sq / 2 + 0.5))
        y = int(np.floor(cy - self.sq / 2 + 0.5))
        x = max(0, min(self.w - self.sq, x))
        y = max(0, min(self.h - self.sq, y))
        return (x, y, self.sq, self.sq)

    def planes(self, t: int):
        """-> (Y [h,w], UV [ceil(h/2), w]) uint8"""
        x, y, s, _ = self.gt_box(t)
        yy = self.bg_y.copy()
        uv = np.full(((self.h + 1) // 2, (self.w + 1) // 2, 2), 128, np.uint8)
        if self.hide is None or not (self.hide[0] <= t < self.hide[1]):
            yy[y:y + s, x:x + s] = self.sq_y
            uv[y // 2:(y + s + 1) // 2, x // 2:(x + s + 1) // 2, 0] = self.sq_u
            uv[y // 2:(y + s + 1) // 2, x // 2:(x + s + 1) // 2, 1] = self.sq_v
        uv = uv.reshape(uv.shape[0], -1)[:, : self.w + (self.w & 1)]
        return yy, uv

    def frame_nv12(self, t: int) -> np.ndarray:
        """packed NV12 buffer (even width/height expected for the packed form)"""
        yy, uv = self.planes(t)
        return np.concatenate([yy.reshape(-1), uv[:, : self.w].reshape(-1)])

    def frame_rgb8(self, t: int) -> np.ndarray:
        """(h,w,3) RGB8 through the reference's integer BT.601 formulas
        (/root/reference/src/nv12_convert.rs:124-126), vectorised"""
        yy, uv = self.planes(t)
        return nv12_planes_to_rgb8(yy, uv[:, : self.w], self.w, self.h)


def nv12_planes_to_rgb8(yy: np.ndarray, uv: np.ndarray, w: int, h: int) -> np.ndarray:
    """Vectorised integer BT.601 limited-range conversion, same arithmetic as
    /root/reference/src/nv12_convert.rs:24-29,124-131 (used to synthesise RGB test frames)."""
    yv = 298 * (yy.astype(np.int32) - 16)
    cols = (np.arange(w) // 2) * 2
    rows = np.arange(h) // 2
    u = uv[rows][:, cols].astype(np.int32) - 128
    v = uv[rows][:, np.minimum(cols + 1, uv.shape[1] - 1)].astype(np.int32) - 128
    r = (yv + 409 * v + 128) >> 8
    g = (yv - 100 * u - 208 * v + 128) >> 8
    b = (yv + 516 * u + 128) >> 8
    return np.clip(np.stack([r, g, b], axis=-1), 0, 255).astype(np.uint8)
