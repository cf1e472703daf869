"""Does the latency of a single tracker depend on WHERE its buffers landed? Creates trackers one after the other in one process
(first keeping the earlier ones alive, then releasing each before the next) and prints every tracker's update latency, twice
(interleaved), so that a per-tracker effect can be told from drift:   python tools/one_tracker_placement.py [trackers] [updates]"""
import sys
import time
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import torch
import gstreamer_vit_tracker_amd as vt
ntrk = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
w, h = 1920, 1080
sc = vt.synth.MovingSquare(w, h, 64, seed=0, path="circle", period=64, amp=3.8 * 64 / (2 * np.pi))
dev = [torch.from_numpy(sc.frame_nv12(t)).cuda() for t in range(64)]
weights = vt.weights.ensure_weights("cfg3")


def make():
    trk = vt.VitTrack.new(weights)
    trk.init_nv12_device(dev[0].data_ptr(), dev[0].data_ptr() + w * h, w, h, w, w, vt.BBox.new(*sc.gt_box(0)))
    return trk


def p50(trk):
    lat = []
    for t in range(1, n + 1):
        p = dev[t % 64].data_ptr()
        a = time.perf_counter()
        r = trk.update_nv12_device(p, p + w * h, w, h, w, w)
        lat.append(time.perf_counter() - a)
        assert r.success
    return float(np.median(np.array(lat[20:]) * 1e3))


alive = [make() for _ in range(ntrk)]
first = [p50(t) for t in alive]
second = [p50(t) for t in alive]
print("all alive, first round : " + "  ".join(f"{v:.4f}" for v in first))
print("all alive, second round: " + "  ".join(f"{v:.4f}" for v in second), flush=True)
del alive
row = []
for _ in range(ntrk):
    t = make()
    row.append((p50(t), p50(t)))
    del t
print("one at a time (2 rounds each): " + "  ".join(f"{a:.4f}/{b:.4f}" for a, b in row))
