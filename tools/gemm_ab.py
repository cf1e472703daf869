"""A/B of GEMM tile configurations in ONE process, interleaved rounds (guide rule 24): prints the
median and minimum microseconds per launch of each configuration on the tracker's four encoder
shapes.  python tools/gemm_ab.py 17,18 30 [rounds]"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [17, 18]
Bs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [30]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
tokens = int(sys.argv[4]) if len(sys.argv) > 4 else 720
D = int(sys.argv[5]) if len(sys.argv) > 5 else 768
epi_names = {1: "resid", 2: "gelu", 4: "qkv"}
for B in Bs:
    M = tokens * B
    for (N, K, epi) in [(3 * D, D, 4), (D, D, 1), (4 * D, D, 2), (D, 4 * D, 1)]:
        t = {c: [] for c in cfgs}
        for r in range(rounds):
            for c in cfgs:
                t[c].append(vt.op_gemm_bench(M, N, K, epi, c, iters=20))
        row = [f"cfg{c}: med {np.median(t[c]):7.1f} min {min(t[c]):7.1f} us {2.0*M*N*K/np.median(t[c])/1e6:5.0f} TF"
               for c in cfgs]
        print(f"B={B:3d} M={M:6d} N={N:4d} K={K:4d} {epi_names[epi]:5s} | " + " | ".join(row), flush=True)
