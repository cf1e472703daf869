"""A/B of GEMM tile configurations on one shape, alternating in one process:
   python tools/gemm_ab.py M N K EPI cfgA,cfgB[,..] [rounds]"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
M, N, K, epi = (int(v) for v in sys.argv[1:5])
cfgs = [int(c) for c in sys.argv[5].split(",")]
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 7
t = {c: [] for c in cfgs}
for r in range(rounds):
    for c in cfgs:
        t[c].append(vt.op_gemm_bench(M, N, K, epi, cfg=c, iters=20))
for c in cfgs:
    us = np.median(t[c])
    print(f"M {M} N {N} K {K} epi {epi} cfg {c}: med {us:7.1f} min {min(t[c]):7.1f} us  {2.0*M*N*K/us/1e6:6.0f} TF", flush=True)
