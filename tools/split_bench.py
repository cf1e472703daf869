"""Throughput of n streams spread over engines of UNEQUAL size on one GPU (the question behind the
tile-count cliff: is 31 = 30 + 1 in two engines cheaper than one engine of 31?).

   python tools/split_bench.py [cfg2|cfg3|cfg5] 31 30+1 30+15 45 30+30+2 [steps >= 200]

Each argument is a '+'-separated list of engine sizes; every engine has its own HIP stream and launch
graph, a step enqueues one pass on every engine and waits for all of them (as bench.py does)."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt

specs = list(sys.argv[1:])
cfg = specs.pop(0) if specs and specs[0].startswith("cfg") else "cfg3"
steps = 60
offset_ms = 0.0       # "off=2.5": engine 0 starts that many ms before the others (phase between engines)
for a in list(specs):
    if a.startswith("off="):
        offset_ms = float(a[4:]); specs.remove(a)
if len(specs) > 1 and specs[-1].isdigit() and int(specs[-1]) >= 200:
    steps = int(specs.pop())
wpath = vt.weights.ensure_weights(cfg)
fw, fh, sq, R = (3840, 2160, 160, 16) if cfg == "cfg5" else (1920, 1080, 64, 16)
sc = vt.synth.MovingSquare(fw, fh, sq, seed=7, path="circle", period=R, amp=3.8 * R / (2 * np.pi))
host = np.stack([sc.frame_nv12(t) for t in range(R)])
clip = torch.from_numpy(host).to("cuda:0")
fbytes, base = host.shape[1], clip.data_ptr()

base_us = None
for spec in specs:
    sizes = [int(s) for s in spec.split("+")]
    n = sum(sizes)
    grps = [vt.Group(wpath, n_streams=b, device=0) for b in sizes]
    frames_at = [[vt.frame_nv12(base + ((t + i) % R) * fbytes, base + ((t + i) % R) * fbytes + fw * fh, fw, fh)
                  for i in range(n)] for t in range(R)]
    off = np.cumsum([0] + sizes)
    for g, grp in enumerate(grps):
        for i in range(sizes[g]):
            grp.init_device(i, frames_at[0][off[g] + i], vt.BBox.new(*sc.gt_box((off[g] + i) % R)))

    def step(t):
        fr = frames_at[t % R]
        for g, grp in enumerate(grps):
            grp.enqueue_device(fr[off[g]:off[g + 1]])

    for t in range(10):
        step(t)
    for grp in grps:
        grp.wait()
    torch.cuda.synchronize()
    if offset_ms > 0 and len(grps) > 1:
        # give engine 0 a head start of offset_ms inside its pass (its queue then stays one pass ahead)
        grps[0].enqueue_device(frames_at[10 % R][off[0]:off[1]])
        time.sleep(offset_ms * 1e-3)
    t0 = time.perf_counter()
    for t in range(10, 10 + steps):
        step(t)
    for grp in grps:
        grp.wait()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok = all(grp.read_state(i)["frames_done"] >= 10 + steps for g, grp in enumerate(grps) for i in range(sizes[g]))
    us = dt / steps / n * 1e6
    print(f"{spec:>12}: {n * steps / dt:8.1f} frames/s  {dt / steps * 1e3:7.3f} ms/step  {us:6.2f} us/frame  ok {ok}",
          flush=True)
    del grps
