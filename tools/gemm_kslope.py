"""time vs K at fixed M, N: slope = per-K-tile cost of the main loop, intercept = prologue + epilogue
usage: python tools/gemm_kslope.py [cfg] [M]"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 17
M = int(sys.argv[2]) if len(sys.argv) > 2 else 21600
for (N, epi) in [(768, 1), (768, 5), (2304, 4), (3072, 2), (3072, 3)]:
    row, pts = [], []
    for K in (128, 256, 768, 1536, 3072, 6144):
        us = min(vt.op_gemm_bench(M, N, K, epi, cfg, iters=20) for _ in range(3))
        pts.append((K, us))
        row.append(f"K={K}: {us:6.1f}us")
    slope = (pts[-1][1] - pts[2][1]) / ((pts[-1][0] - pts[2][0]) / 64)
    icpt = pts[2][1] - slope * pts[2][0] / 64
    tiles = ((M + 255) // 256) * (N // 256)
    print(f"M={M} N={N} epi={epi} cfg={cfg} | " + "  ".join(row) +
          f" | {slope:.3f} us/K-tile over {tiles} tiles, intercept {icpt:.1f} us", flush=True)
