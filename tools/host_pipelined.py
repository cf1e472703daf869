"""Pipelined host ingest (vt_group_enqueue_host / vt_group_wait_next) for a rocprofv3 trace:
   rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python3 tools/host_pipelined.py 30 1 12
usage: python tools/host_pipelined.py [streams_per_engine] [engines] [steps] [sync|pipe] [eager|graph] [distinct]
(distinct: every stream reads its own frame of the clip, as bench.py's leg does, instead of all the same one)"""
import sys, time, threading
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt

B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
G = int(sys.argv[2]) if len(sys.argv) > 2 else 2
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
sync = len(sys.argv) > 4 and sys.argv[4] == "sync"
w, h, R = 1920, 1080, 16
wts = vt.weights.ensure_weights("cfg3")
sc = vt.synth.MovingSquare(w, h, 64, seed=0, path="circle", period=R, amp=3.8 * R / (2 * np.pi))   # closed path, as bench.py
clip = [vt.NV12Frame(sc.frame_nv12(t), w, h) for t in range(R)]
eager = len(sys.argv) > 5 and sys.argv[5] == "eager"
groups = [vt.Group(wts, n_streams=B, use_graph=not eager) for _ in range(G)]
distinct = len(sys.argv) > 6 and sys.argv[6] == "distinct"
ph = [i % R if distinct else 0 for i in range(B)]
for g in groups:
    for i in range(B):
        g.init_host(i, clip[ph[i]], vt.BBox.new(*sc.gt_box(ph[i])))


def fr(t):
    return [clip[(t + ph[i]) % R] for i in range(B)]


def run(g, n, out):
    ok = True
    if sync:
        for t in range(1, n + 1):
            ok = ok and all(r.success for r in g.update_host(fr(t)))
    else:
        g.enqueue_host(fr(1))
        for t in range(2, n + 1):
            g.enqueue_host(fr(t))
            ok = ok and all(r.success for r in g.wait_next())
        ok = ok and all(r.success for r in g.wait_next())
    out.append(ok)


oks = []
t0 = time.perf_counter()
th = [threading.Thread(target=run, args=(g, steps, oks)) for g in groups]
[x.start() for x in th]
[x.join() for x in th]
dt = time.perf_counter() - t0
print(f"{'synchronous' if sync else 'pipelined'}{' eager' if eager else ''} host frames, {G} engines x {B} streams: {G * B * steps / dt:.0f} "
      f"tracked frames/s ({dt / steps * 1e3:.2f} ms per step of {G * B} frames), all tracked: {all(oks)}, "
      f"redone passes: {sum(g.host_redos() for g in groups)}")
