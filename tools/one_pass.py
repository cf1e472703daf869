"""N passes of one 30-stream cfg3 engine on device-resident 1080p frames, for profilers (per-kernel PMC rows of a real pass):
python tools/one_pass.py [streams] [passes]"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import numpy as np
import torch
import gstreamer_vit_tracker_amd as vt
B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
w, h = 1920, 1080
sc = vt.synth.MovingSquare(w, h, 64, seed=0, path="circle", period=64, amp=3.8 * 64 / (2 * np.pi))
dev = [torch.from_numpy(sc.frame_nv12(t)).cuda() for t in range(8)]
fr = [vt.frame_nv12(d.data_ptr(), d.data_ptr() + w * h, w, h) for d in dev]
g = vt.Group(vt.weights.ensure_weights("cfg3"), n_streams=B)
for i in range(B):
    g.init_device(i, fr[0], vt.BBox.new(*sc.gt_box(0)))
for t in range(1, n + 1):
    r = g.update_device([fr[t % 8]] * B)
print("ok", all(x.success for x in r))
