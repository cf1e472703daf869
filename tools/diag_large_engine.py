"""Where does an engine of many streams leave the 30-stream engine? Same two frames into every stream of
both engines, taps on; per stage, the relative difference of stream s of the big engine to stream 0 of the
small one (identical inputs: only the kernels' tile shapes / tile order may differ).
    python3 tools/diag_large_engine.py [B ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gstreamer_vit_tracker_amd as gpu            # noqa: E402
from gstreamer_vit_tracker_amd import weights      # noqa: E402


def rel(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [330]
    path = weights.ensure_weights("cfg3")
    w, h = 1920, 1080
    sc = gpu.synth.MovingSquare(w, h, 64, seed=411)
    f0, f1 = gpu.NV12Frame(sc.frame_nv12(0), w, h), gpu.NV12Frame(sc.frame_nv12(1), w, h)
    box = gpu.BBox.new(*sc.gt_box(0))
    small = gpu.Group(path, n_streams=30)
    small.enable_taps(True)
    for i in range(30):
        small.init_host(i, f0, box)
    mi = small.model_info()
    names = ["patches", "tokens0"] + [f"layer{l}" for l in range(mi.layers)] + ["feat", "head_out"]
    ref = {}
    for k, f in enumerate((f0, f1)):
        rs = small.update_host([f] * 30)
        ref[k] = ({n: small.read_tensor(n, 0).copy() for n in names}, rs[0], small.read_state(0)["last_fbox"].copy())
        print(f"[30] frame {k}: {rs[0]} fbox {ref[k][2]}")
        for n in names:
            d = rel(small.read_tensor(n, 29), ref[k][0][n])
            if d:
                print(f"   30-stream engine: stream 29 differs from stream 0 at {n}: {d:.3e}")
    for B in sizes:
        big = gpu.Group(path, n_streams=B)
        big.enable_taps(True)
        for i in range(B):
            big.init_host(i, f0, box)
        for k, f in enumerate((f0, f1)):
            rb = big.update_host([f] * B)
            print(f"[{B}] frame {k}: {rb[0]} fbox {big.read_state(0)['last_fbox']}")
            for s in sorted({0, B // 2, B - 1}):
                print(f"   stream {s}: " + "  ".join(f"{n}={rel(big.read_tensor(n, s), ref[k][0][n]):.2e}" for n in names))
        big.close()


if __name__ == "__main__":
    main()
