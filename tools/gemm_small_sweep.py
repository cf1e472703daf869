"""tile-configuration sweep of the 4-wave GEMM kernel on the shapes of 1, 2 and 4 streams (one process,
interleaved rounds): python tools/gemm_small_sweep.py [tokens] [D]"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
tokens = int(sys.argv[1]) if len(sys.argv) > 1 else 720
D = int(sys.argv[2]) if len(sys.argv) > 2 else 768
EPI = {"xresid": 1, "gelu": 2, "qkv": 4}
for streams in (1, 2, 4):
    M = tokens * streams
    for name, N, K in (("qkv", 3 * D, D), ("xresid", D, D), ("gelu", 4 * D, D), ("xresid", D, 4 * D)):
        res = {}
        for r in range(3):
            for cfg in (0, 1, 2, 3, 4, 5, 6):
                try:
                    res.setdefault(cfg, []).append(vt.op_gemm_bench(M, N, K, EPI[name], cfg=cfg, iters=30))
                except Exception:
                    pass
        auto = vt.op_gemm_bench(M, N, K, EPI[name], cfg=-1, iters=30)
        best = min(res, key=lambda c: np.median(res[c]))
        print(f"M {M:5d} {name:7s} N {N:5d} K {K:5d}: " + "  ".join(f"{c}:{np.median(v):6.1f}" for c, v in res.items()) +
              f"   auto {auto:6.1f}  best cfg {best}", flush=True)
