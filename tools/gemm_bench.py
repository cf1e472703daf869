"""GEMM tile-configuration sweep on the tracker's shapes (run on the GPU box)."""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
names = {0: "64x64x4", 1: "128x128x3", 2: "64x64x2", 3: "128x128x2", 18: "256x256", 19: "256x256 persistent"}
epi_names = {1: "resid", 2: "gelu", 4: "qkv"}
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [2, 3, 18]
Bs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 4, 8, 16, 32]
for B in Bs:
    M = 720 * B
    for (N, K, epi) in [(2304, 768, 4), (768, 768, 1), (3072, 768, 2), (768, 3072, 1)]:
        row = []
        for cfg in cfgs:
            if cfg >= 18 and N % 256:
                continue
            us = vt.op_gemm_bench(M, N, K, epi, cfg, iters=30)
            row.append(f"{names[cfg]} {us:7.1f}us {2.0*M*N*K/us/1e6:5.0f}TF")
        print(f"B={B:2d} M={M:5d} N={N:4d} K={K:4d} {epi_names[epi]:5s} | " + " | ".join(row), flush=True)
