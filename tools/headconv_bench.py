"""The head's band kernel (k_head.hip): time per launch over band heights R and column groups, at one stream and at
the benchmark's batch (run on the GPU box). Usage: python tools/headconv_bench.py [B,B,...] [grid] [C] [D]"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt

Bs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [1, 2, 8, 30]
grid = int(sys.argv[2]) if len(sys.argv) > 2 else 24
C = int(sys.argv[3]) if len(sys.argv) > 3 else 128
D = int(sys.argv[4]) if len(sys.argv) > 4 else 768
for B in Bs:
    M = B * grid * grid
    for conv, K, name in ((True, 9 * C, "conv3x3"), (False, D, "conv1x1")):
        fl = 2.0 * M * C * K
        row = []
        us = vt.op_headconv_bench(B, grid, C if conv else D, C, conv, 0, 0, iters=50)
        row.append(f"plan {us:6.1f}us {fl / us / 1e6:5.0f}TF")
        for R in range(1, grid + 1):
            if (R * grid + 15) // 16 > 7:
                break
            for ncb in ((2, 1) if C == 128 else (1,)):
                if B * ((grid + R - 1) // R) * (2 // ncb if C == 128 else 1) > 1024:
                    continue
                us = vt.op_headconv_bench(B, grid, C if conv else D, C, conv, R, ncb, iters=50)
                row.append(f"R{R}n{ncb} {us:6.1f}")
        print(f"B={B:2d} {name} M={M:5d} K={K:4d} | " + " | ".join(row), flush=True)
    # the 1x1 layer with the final LayerNorm in front: two launches (LayerNorm kernel + band kernel) against one
    if D in (768, 1024):
        ntok, off = grid * grid + (grid // 2) ** 2, (grid // 2) ** 2
        row = [f"two launches {vt.op_headconv_ln_bench(B, grid, D, C, ntok, off, fused=False):6.1f}us",
               f"one (plan) {vt.op_headconv_ln_bench(B, grid, D, C, ntok, off, fused=True):6.1f}us"]
        for R in range(1, grid + 1):
            for ncb in ((2, 1) if C == 128 else (1,)):
                try:
                    us = vt.op_headconv_ln_bench(B, grid, D, C, ntok, off, fused=True, R=R, ncb=ncb)
                except vt.VtError:
                    continue
                row.append(f"R{R}n{ncb} {us:6.1f}")
        print(f"B={B:2d} ln+conv1x1 D={D} | " + " | ".join(row), flush=True)
