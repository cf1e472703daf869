import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
modes = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 3]
Bs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16, 32]
N, H = 720, 12
for B in Bs:
    row = []
    for m in modes:
        us = vt.op_attention_bench(B, N, H, m, iters=20)
        row.append(f"mode{m} {us:7.1f}us {4.0*B*N*N*H*64/us/1e6:5.0f}TF")
    print(f"B={B:2d} | " + " | ".join(row), flush=True)
