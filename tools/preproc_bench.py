"""Crop / resize kernel time by target size (rectangles that do not fit the tile kernel's LDS buffer).
python tools/preproc_bench.py cfg3|cfg2 [streams]   (run on the GPU box; VITTRACK_HIP_LIB selects the build)"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import numpy as np
import gstreamer_vit_tracker_amd as vt

cfg = {"cfg3": "vitb16_t192_s384", "cfg2": "vitb16_t128_s256"}[sys.argv[1] if len(sys.argv) > 1 else "cfg3"]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 30
w, h = 1920, 1080
weights = vt.weights.ensure_weights(cfg)
import torch
for sq in (48, 64, 96, 128, 160, 200, 256):
    sc = vt.synth.MovingSquare(w, h, sq, seed=1)
    buf = torch.from_numpy(sc.frame_nv12(0)).cuda()
    fr = vt.frame_nv12(buf.data_ptr(), buf.data_ptr() + w * h, w, h)
    grp = vt.Group(weights, n_streams=B)
    for i in range(B):
        grp.init_device(i, fr, vt.BBox.new(*sc.gt_box(0)))
    us = []
    for _ in range(3):
        for i in range(B):
            grp.set_state_box(i, [float(v) for v in sc.gt_box(0)])     # the crop is cut around THIS box every time
        prof = grp.profile_device([fr] * B, iters=1)
        us.append([p["ms"] * 1e3 for p in prof if p["name"] == "preproc_search"][0])
    print(f"{cfg} {B} streams, target {sq:3d} px (crop side {4 * sq} px): preproc {np.median(us):6.1f} us (min {min(us):6.1f})", flush=True)
    del grp
