"""PCIe-inclusive throughput of the batched host-frame path (vt_group_update_host): B streams per
engine, 1080p NV12 frames in ordinary host memory; only the search windows are packed and copied.
usage: python tools/host_throughput.py [streams_per_engine] [engines] [steps]"""
import sys, time, threading
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
w, h, R = 1920, 1080, 16
wts = vt.weights.ensure_weights("cfg3")
sc = vt.synth.MovingSquare(w, h, 64, seed=0)
clip = [vt.NV12Frame(sc.frame_nv12(t), w, h) for t in range(R)]
groups = [vt.Group(wts, n_streams=B) for _ in range(G)]
for g in groups:
    for i in range(B):
        g.init_host(i, clip[0], vt.BBox.new(*sc.gt_box(0)))


def run(g, n, out):
    ok = True
    for t in range(n):
        res = g.update_host([clip[(t + 1) % R]] * B)
        ok = ok and all(r.success for r in res)
    out.append(ok)


for g in groups:
    run(g, 3, [])
oks = []
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(g, steps, oks)) for g in groups]
[x.start() for x in th]
[x.join() for x in th]
dt = time.perf_counter() - t0
print(f"host frames, {G} engines x {B} streams: {G * B * steps / dt:.0f} tracked frames/s "
      f"({dt / steps * 1e3:.2f} ms per step of {G * B} frames), all tracked: {all(oks)}")
