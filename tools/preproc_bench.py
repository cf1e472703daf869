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

# mixed target sizes in ONE engine: the tier of a pass is the largest any stream needs, so one large target puts every stream's
# tiles on the big buffer (2 blocks per CU at 64 KiB instead of 8 at 16 KiB)
for big in (160, 200, 256):
    scs = [vt.synth.MovingSquare(w, h, 64, seed=1)] * (B - 1) + [vt.synth.MovingSquare(w, h, big, seed=1)]
    bufs = {id(s): torch.from_numpy(s.frame_nv12(0)).cuda() for s in scs}
    frs = [vt.frame_nv12(bufs[id(s)].data_ptr(), bufs[id(s)].data_ptr() + w * h, w, h) for s in scs]
    grp = vt.Group(weights, n_streams=B)
    for i in range(B):
        grp.init_device(i, frs[i], vt.BBox.new(*scs[i].gt_box(0)))
    us = []
    for _ in range(3):
        for i in range(B):
            grp.set_state_box(i, [float(v) for v in scs[i].gt_box(0)])
        prof = grp.profile_device(frs, iters=1)
        us.append([p["ms"] * 1e3 for p in prof if p["name"] == "preproc_search"][0])
    print(f"{cfg} {B} streams, {B - 1} targets of 64 px + ONE of {big} px: preproc {np.median(us):6.1f} us (min {min(us):6.1f})", flush=True)
    del grp
