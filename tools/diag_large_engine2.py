"""final results of an engine of B streams under graph replay / eager launches, against the 30-stream engine
    python3 tools/diag_large_engine2.py B [graph|eager] [bigfirst]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gstreamer_vit_tracker_amd as gpu            # noqa: E402
from gstreamer_vit_tracker_amd import weights      # noqa: E402

B = int(sys.argv[1])
graph = "eager" not in sys.argv
path = weights.ensure_weights("cfg3")
w, h = 1920, 1080
sc = gpu.synth.MovingSquare(w, h, 64, seed=411)
f0, f1 = gpu.NV12Frame(sc.frame_nv12(0), w, h), gpu.NV12Frame(sc.frame_nv12(1), w, h)
box = gpu.BBox.new(*sc.gt_box(0))
if "bigfirst" in sys.argv:
    big = gpu.Group(path, n_streams=B, use_graph=graph)
    small = gpu.Group(path, n_streams=30)
else:
    small = gpu.Group(path, n_streams=30)
    big = gpu.Group(path, n_streams=B, use_graph=graph)
for i in range(B):
    big.init_host(i, f0, box)
for i in range(30):
    small.init_host(i, f0, box)
for k, f in enumerate((f0, f1, f1, f0)):
    rb, rs = big.update_host([f] * B), small.update_host([f] * 30)
    fb = np.array([big.read_state(i)["last_fbox"] for i in range(B)])
    same = sum(1 for r in rb if r.bbox == rb[0].bbox and r.score == rb[0].score)
    odd = [i for i, r in enumerate(rb) if not (r.bbox == rs[0].bbox and r.score == rs[0].score)]
    print(f"[{B} {'graph' if graph else 'eager'}] frame {k}: big[0] {rb[0]} small[0] {rs[0]}; {same} of {B} equal big[0]; "
          f"{len(odd)} differ from small[0] (first {odd[:8]}); fbox spread {np.ptp(fb, axis=0)}")
