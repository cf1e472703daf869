import sys
sys.path.insert(0, '.')
import gstreamer_vit_tracker_amd as vt
for (M, N, epi, cfg) in [(11520, 2304, 4, 3), (11520, 3072, 2, 3), (11520, 768, 1, 3), (11520, 768, 1, 2), (5760, 2304, 4, 3)]:
    row = []
    for K in (128, 256, 768, 1536, 3072):
        us = vt.op_gemm_bench(M, N, K, epi, cfg, iters=20)
        row.append(f"K={K}: {us:6.1f}us")
    print(f"M={M} N={N} epi={epi} cfg={cfg} | " + "  ".join(row), flush=True)
