"""Per-layer relative deviation of the HIP residual stream from the oracle's (one frame), with the
oracle's softmax reference point at the true row maximum (the spec) and at 0 (what the HIP kernel
uses inside its +-32 window): python tools/diag_taps.py cfg5"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
from oracle import vit_ref as R

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w, h, sq = (3840, 2160, 160) if cfg == "cfg5" else (1920, 1080, 64)
weights = vt.weights.ensure_weights(cfg)
sc = vt.synth.MovingSquare(w, h, sq, seed=3)
buf = sc.frame_nv12(0)
trk = vt.VitTrack(weights)
g = trk.as_group()
g.enable_taps(True)
f = vt.NV12Frame(buf, w, h)
trk.init(f, vt.BBox.new(*sc.gt_box(0)))
r = trk.update(f)
mi = trk.model_info()
n, d = mi.tokens_template + mi.tokens_search, mi.dim
for mode in ("max", "zero"):
    R.SOFTMAX_REF = mode
    ref = R.VitTrackRef(weights)
    of = R.Frame.nv12(buf, w, h)
    ref.init(of, sc.gt_box(0))
    rr = ref.update(of, taps=True)
    errs = []
    for l in range(mi.layers):
        x = g.read_tensor(f"layer{l}").reshape(n, d)
        e = ref.last[f"layer{l}"]
        errs.append(float(np.abs(x - e).max() / np.abs(e).max()))
    ho = g.read_tensor("head_out").reshape(mi.tokens_search, 8)[:, :5]
    eh = np.abs(ho - ref.last["head_out"][:, :5]).max()
    rms = float(np.sqrt(np.mean((g.read_tensor(f"layer{mi.layers-1}").reshape(n, d) - ref.last[f"layer{mi.layers-1}"]) ** 2)) /
                np.sqrt(np.mean(ref.last[f"layer{mi.layers-1}"] ** 2)))
    print(f"[{cfg}] oracle softmax ref={mode}: layer max-rel err first {errs[0]:.2e} mid {errs[len(errs)//2]:.2e} "
          f"last {errs[-1]:.2e}; last-layer rms-rel {rms:.2e}; head logits max abs diff {eh:.4f}; "
          f"boxes hip {r.bbox} oracle {rr.bbox} score {r.score:.4f} / {rr.score:.4f}", flush=True)
