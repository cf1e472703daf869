"""per-frame differences of a HIP closed loop against a committed oracle trajectory: python tools/diag_traj2.py traj_cfg5_300.npz"""
import sys, os
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
fx = dict(np.load(os.path.join("tests", "golden", sys.argv[1])))
cfg = str(fx["config"])
w, h, n = int(fx["frame_w"]), int(fx["frame_h"]), int(fx["frames"])
sc = vt.synth.MovingSquare(w, h, int(fx["square"]), seed=int(fx["seed"]))
trk = vt.VitTrack(vt.weights.ensure_weights(cfg))
g = trk.as_group()
rows = []
for t in range(n):
    f = vt.NV12Frame(sc.frame_nv12(t), w, h)
    if t == 0:
        trk.init(f, vt.BBox.new(*sc.gt_box(0)))
    r = trk.update(f)
    st = g.read_state()
    rows.append((t, r.bbox, st["last_fbox"].copy(), st["last_idx"]))
d = np.array([np.array(r[1]) - fx["bbox"][r[0]] for r in rows])
print("max |delta| per coordinate (x, y, w, h):", np.abs(d).max(axis=0), "frames with any delta:", int((np.abs(d).max(axis=1) > 0).sum()))
for t, box, fb, idx in rows:
    dd = np.array(box) - fx["bbox"][t]
    if np.abs(dd).max() >= 2 or (t > 0 and np.abs(np.array(rows[t-1][1]) - fx["bbox"][t-1]).max() >= 2):
        print(f"frame {t:3d}: hip {box} fbox {np.round(fb, 2)} oracle {fx['bbox'][t].tolist()} delta {dd.tolist()} idx {idx}/{int(fx['idx'][t])} gt {fx['gt'][t].tolist()} margin {float(fx['margin'][t]):.3f}")
