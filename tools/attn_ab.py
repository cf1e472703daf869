"""A/B of attention kernel modes in one process: python tools/attn_ab.py 3,4,5,6 30 [rounds]"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
modes = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [3]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 30
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
N, H = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (720, 12)
t = {m: [] for m in modes}
for r in range(rounds):
    for m in modes:
        t[m].append(vt.op_attention_bench(B, N, H, m, iters=20))
for m in modes:
    us = np.median(t[m])
    print(f"mode {m}: med {us:7.1f} min {min(t[m]):7.1f} us  {4.0*B*N*N*H*64/us/1e6:6.0f} TF", flush=True)
