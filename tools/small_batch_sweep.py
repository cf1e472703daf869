"""tile-configuration sweep of the kernel of k_gemm.hip on the shapes of 1-3 streams (run on the GPU box):
python tools/small_batch_sweep.py [streams ...]   ->   us per launch for every configuration 0..8 (an XCD's run of workgroups: row panels x all columns / column tiles x all rows) and the picker's choice"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
Bs = [int(v) for v in sys.argv[1:]] or [1, 2, 3]
for B in Bs:
    M = 720 * B
    for (name, N, K, epi) in [("qkv", 2304, 768, 4), ("xresid", 768, 768, 1), ("gelu", 3072, 768, 2), ("xresid", 768, 3072, 1)]:
        row = []
        for cfg in range(9):
            try:
                us = [min(vt.op_gemm_bench(M, N, K, epi, cfg | o, iters=200) for _ in range(3)) for o in (0x100, 0x200)]
                row.append(f"{cfg}: " + "/".join(f"{u:4.1f}" for u in us))
            except Exception:
                row.append(f"{cfg}:   -  ")
        auto = min(vt.op_gemm_bench(M, N, K, epi, -1, iters=200) for _ in range(3))
        print(f"M {M:5d} {name:7s} N {N:5d} K {K:5d}: " + "  ".join(row) + f"   auto {auto:5.1f}", flush=True)
