"""Regenerates the committed fixtures under tests/golden/.

  nv12_kat.json     known-answer vectors for the reference's NV12->RGB8 conversion, computed here
                    with plain Python integers straight from the formulas at
                    /root/reference/src/nv12_convert.rs:24-29 (tables) and :124-131 (per pixel),
                    :111-113/:152 (UV addressing) — independent of oracle/vt_oracle.c, which the
                    tests then check against these vectors. (The reference ships no vectors.)
  tiny_forward.npz  the oracle's outputs for one fixed frame on the tiny model: pins the oracle
                    against accidental change and gives the GPU tests a fixture that does not need
                    the oracle at run time. Build-defined model => PARITY UNPINNED vs the reference.

Run:  python tests/golden/make_golden.py
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)


def px(y, u, v):
    """src/nv12_convert.rs:24-29,124-131 with Python ints (>> is an arithmetic shift)"""
    yv = 298 * (y - 16)
    r = (yv + 409 * (v - 128) + 128) >> 8
    g = (yv - 100 * (u - 128) - 208 * (v - 128) + 128) >> 8
    b = (yv + 516 * (u - 128) + 128) >> 8
    return [min(max(c, 0), 255) for c in (r, g, b)]


def frame(buf, w, h):
    """src/nv12_convert.rs:46-169 with stride == width, UV row = row // 2, UV pair of column
    col & ~1 (for odd w the last pixel's V is the byte after the row, :152-154)"""
    out = []
    for row in range(h):
        line = []
        for col in range(w):
            uv = w * h + (row // 2) * w + (col & ~1)
            line.append(px(buf[row * w + col], buf[uv], buf[uv + 1]))
        out.append(line)
    return out


def main():
    kat = {"pixels": [], "frames": []}
    for yuv in [(16, 128, 128), (235, 128, 128), (81, 90, 240), (0, 0, 0), (255, 255, 255),
                (145, 54, 34), (41, 240, 110), (128, 128, 128), (16, 0, 255), (235, 255, 0),
                (17, 128, 128), (234, 129, 127)]:
        kat["pixels"].append({"yuv": list(yuv), "rgb": px(*yuv)})
    rng = np.random.default_rng(20240)
    for (w, h) in [(6, 4), (8, 6), (5, 3), (7, 4), (4, 5)]:
        n = w * h + ((h + 1) // 2) * w + 1
        buf = [int(v) for v in rng.integers(0, 256, n)]
        kat["frames"].append({"w": w, "h": h, "nv12": buf, "rgb": frame(buf, w, h)})
    with open(os.path.join(HERE, "nv12_kat.json"), "w") as f:
        json.dump(kat, f)
    print("wrote nv12_kat.json:", len(kat["pixels"]), "pixels,", len(kat["frames"]), "frames")

    import gstreamer_vit_tracker_amd as vt
    from oracle import vit_ref as R
    wts = vt.weights.ensure_weights("tiny", force=True)
    sc = vt.synth.MovingSquare(640, 480, 64, seed=7)
    trk = R.VitTrackRef(wts)
    buf = sc.frame_nv12(5)
    fr = R.Frame.nv12(buf, 640, 480)
    box = sc.gt_box(5)
    trk.init(fr, box)
    res = trk.update(fr, taps=True)
    out = trk.last
    np.savez_compressed(
        os.path.join(HERE, "tiny_forward.npz"),
        scene=np.array([640, 480, 64, 7, 5]), init_box=np.array(box),
        patches_sha256=np.frombuffer(hashlib.sha256(out["patches"].tobytes()).digest(), np.uint8),
        patches_rows=out["patches"][[0, 15, 16, 47, 79]],
        tokens0=out["tokens0"].astype(np.float32), layer1=out["layer1"].astype(np.float32),
        feat=out["feat"].astype(np.float32), head_out=out["head_out"].astype(np.float32),
        geo=out["geo"], score=np.float32(res.score), bbox=np.array(res.bbox),
        fbox=res.fbox.astype(np.float32),
        weights_sha256=np.frombuffer(hashlib.sha256(open(wts, "rb").read()).digest(), np.uint8))
    print("wrote tiny_forward.npz:", res)


if __name__ == "__main__":
    main()
