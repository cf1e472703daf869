"""Does a pinned host->device copy on its own stream run WHILE an engine pass occupies the GPU?
Issues the copy ~1 ms into a pass of 30 streams (4.9 ms) and reports when it completed relative to
its issue and to the pass:  python tools/copy_overlap_probe.py [eager]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt

eager = len(sys.argv) > 1 and sys.argv[1] == "eager"
B, fw, fh, R = 30, 1920, 1080, 8
w = vt.weights.ensure_weights("cfg3")
sc = vt.synth.MovingSquare(fw, fh, 64, seed=3, path="circle", period=R, amp=3.8 * R / (2 * np.pi))
host = np.stack([sc.frame_nv12(t) for t in range(R)])
clip = torch.from_numpy(host).to("cuda:0")
fb, base = host.shape[1], clip.data_ptr()
frames = [[vt.frame_nv12(base + ((t + i) % R) * fb, base + ((t + i) % R) * fb + fw * fh, fw, fh) for i in range(B)]
          for t in range(R)]
g = vt.Group(w, n_streams=B, device=0, use_graph=not eager)
for i in range(B):
    g.init_device(i, frames[0][i], vt.BBox.new(*sc.gt_box(i % R)))
for t in range(5):
    g.enqueue_device(frames[t % R]); g.wait()
side = torch.cuda.Stream()
for mb in (1, 9, 64):
    src = torch.empty(mb << 20, dtype=torch.uint8).pin_memory()
    dst = torch.empty(mb << 20, dtype=torch.uint8, device="cuda:0")
    with torch.cuda.stream(side):          # idle GPU: the copy's own time
        dst.copy_(src, non_blocking=True)
    side.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        dst.copy_(src, non_blocking=True)
    side.synchronize()
    alone = time.perf_counter() - t0
    rows = []
    for rep in range(5):
        p0 = time.perf_counter()
        g.enqueue_device(frames[rep % R])
        time.sleep(0.001)
        c0 = time.perf_counter()
        with torch.cuda.stream(side):
            dst.copy_(src, non_blocking=True)
        side.synchronize()
        c1 = time.perf_counter()
        g.wait()
        p1 = time.perf_counter()
        rows.append((c0 - p0, c1 - c0, p1 - p0))
    med = np.median(np.array(rows), axis=0) * 1e3
    print(f"{mb:3d} MiB pinned H2D: alone {alone * 1e3:.3f} ms; issued {med[0]:.2f} ms into a pass of {med[2]:.2f} ms "
          f"it completed {med[1]:.3f} ms after its issue ({'overlapped' if med[1] < 0.6 * (med[2] - med[0]) else 'waited for the pass'})",
          flush=True)
