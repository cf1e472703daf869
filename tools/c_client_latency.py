"""The single-tracker update latency as a COMPILED host sees it (harness/c_client, plain C99 over dlopen; the reference host is
Rust, src/pipeline.rs:109-120): writes a 1080p synthetic clip, runs `c_client run` on cfg3 and prints its stderr latency line."""
import os, subprocess, sys, tempfile
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
w, h, n = 1920, 1080, int(sys.argv[1]) if len(sys.argv) > 1 else 320
sc = vt.synth.MovingSquare(w, h, 64, seed=0, path="circle", period=64, amp=3.8 * 64 / (2 * np.pi))
with tempfile.TemporaryDirectory() as d:
    clip = os.path.join(d, "clip.nv12")
    with open(clip, "wb") as f:
        for t in range(n):
            f.write(sc.frame_nv12(t % 64).tobytes())
    out = subprocess.run([os.path.join(root, "harness", "c_client"), "run", vt.LIB_PATH, vt.weights.ensure_weights("cfg3"), clip,
                          str(w), str(h), str(n)] + [str(int(v)) for v in sc.gt_box(0)], capture_output=True, text=True)
    ok = sum(int(l.split()[1]) for l in out.stdout.strip().splitlines())
    print(out.stderr.strip().splitlines()[-1], f"({ok} of {n} updates successful)")
