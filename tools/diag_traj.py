"""per-frame float boxes of the HIP path and the oracle on the same clip (diagnostic)"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
from oracle import vit_ref as R
cfg, frames, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w, h = 1920, 1080
wts = vt.weights.ensure_weights(cfg)
sc = vt.synth.MovingSquare(w, h, 64, seed=seed)
trk, ref = vt.VitTrack(wts), R.VitTrackRef(wts)
g = trk.as_group()
for t in range(frames):
    buf = sc.frame_nv12(t)
    f, of = vt.NV12Frame(buf, w, h), R.Frame.nv12(buf, w, h)
    if t == 0:
        trk.init(f, vt.BBox.new(*sc.gt_box(0))); ref.init(of, sc.gt_box(0))
    rg, rr = trk.update(f), ref.update(of, taps=True)
    st = g.read_state(0)
    ho = g.read_tensor("head_out").reshape(-1, 8)
    dh = np.abs(ho[:, :5] - ref.last["head_out"][:, :5]).max()
    d = np.abs(st["last_fbox"] - rr.fbox)
    flag = " <<<" if max(abs(a - b) for a, b in zip(rg.bbox, rr.bbox)) > 1 else ""
    print(f"t={t:3d} gpu {np.round(st['last_fbox'],2)} ref {np.round(rr.fbox,2)} |d| {np.round(d,3)} idx {st['last_idx']}/{rr.idx} score {rg.score:.4f}/{rr.score:.4f} max|dlogit| {dh:.3f}{flag}")
