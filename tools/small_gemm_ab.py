"""small-M and head GEMM shapes across the 4-wave kernel's configurations (one process)"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
shapes = [(720, 2304, 768, 4, "qkv B=1"), (720, 768, 768, 1, "proj B=1"), (720, 3072, 768, 2, "fc1 B=1"),
          (720, 768, 3072, 1, "fc2 B=1"), (17280, 128, 1152, 3, "head conv B=30"), (17280, 128, 768, 3, "head 1x1 B=30"),
          (576, 128, 1152, 3, "head conv B=1")]
for (M, N, K, epi, name) in shapes:
    row = []
    for c in (0, 1, 2, 3):
        if c in (1, 3) and N % 128: continue
        try:
            t = [vt.op_gemm_bench(M, N, K, epi, c, iters=30) for _ in range(3)]
            row.append(f"cfg{c} {np.median(t):6.1f}us")
        except Exception as e:
            row.append(f"cfg{c} n/a")
    print(f"{name:16s} M={M:6d} N={N:4d} K={K:4d} | " + " | ".join(row), flush=True)
