"""One tracker, N synchronous updates on 1080p NV12 frames (the reference's own call pattern, src/pipeline.rs:55,109-120), for
profilers: python tools/one_tracker.py [updates] [host|device]    prints p50 / p99 of the update latency"""
import sys
import time
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import torch
import gstreamer_vit_tracker_amd as vt
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
where = sys.argv[2] if len(sys.argv) > 2 else "device"
w, h = 1920, 1080
sc = vt.synth.MovingSquare(w, h, 64, seed=0, path="circle", period=64, amp=3.8 * 64 / (2 * np.pi))
frames = [sc.frame_nv12(t) for t in range(64)]
dev = [torch.from_numpy(f).cuda() for f in frames]
trk = vt.VitTrack.new(vt.weights.ensure_weights("cfg3"))
if where == "host":
    trk.init(vt.NV12Frame(frames[0], w, h), vt.BBox.new(*sc.gt_box(0)))
else:
    trk.init_nv12_device(dev[0].data_ptr(), dev[0].data_ptr() + w * h, w, h, w, w, vt.BBox.new(*sc.gt_box(0)))
lat = []
for t in range(1, n + 1):
    a = time.perf_counter()
    if where == "host":
        r = trk.update(vt.NV12Frame(frames[t % 64], w, h))
    else:
        p = dev[t % 64].data_ptr()
        r = trk.update_nv12_device(p, p + w * h, w, h, w, w)
    lat.append(time.perf_counter() - a)
    assert r.success
lat = np.array(lat[20:]) * 1e3
print(f"{where} pointer, {n} updates: p50 {np.median(lat):.4f} ms  p99 {np.percentile(lat, 99):.4f} ms  {1e3 / lat.mean():.1f} updates/s")
