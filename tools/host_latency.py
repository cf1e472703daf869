"""Latency of the literal drop-in calls with HOST frames (the reference's call pattern):
update(&ArrayView3<u8>) on a 1080p RGB8 frame and the fused NV12 form."""
import sys, time
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
w, h = 1920, 1080
wts = vt.weights.ensure_weights(cfg)
sc = vt.synth.MovingSquare(w, h, 64, seed=0)
frames_nv12 = [vt.NV12Frame(sc.frame_nv12(t), w, h) for t in range(40)]
frames_rgb = [sc.frame_rgb8(t) for t in range(40)]
for name, frames in (("nv12 host", frames_nv12), ("rgb8 host", frames_rgb)):
    trk = vt.VitTrack.new(wts)
    trk.init(frames[0], vt.BBox.new(*sc.gt_box(0)))
    lat = []
    for rep in range(3):
        for t in range(40):
            a = time.perf_counter()
            r = trk.update(frames[t])
            lat.append(time.perf_counter() - a)
    lat = np.array(lat[20:]) * 1e3
    print(f"{cfg} {name}: update p50 {np.median(lat):.3f} ms  p99 {np.percentile(lat, 99):.3f} ms  -> {1e3/np.median(lat):.0f} fps; last {r}")
