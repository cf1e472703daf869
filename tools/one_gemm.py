"""one GEMM shape on device-resident random operands, for profilers: python tools/one_gemm.py M N K EPI CFG ITERS
(EPI: the library's numbering - 0 xpos, 1 xresid, 2 gelu, 3 relu, 4 qkv, 5 x; CFG < 0: the launcher's choice)"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
M, N, K, epi, cfg, iters = (int(v) for v in sys.argv[1:7])
print(f"{vt.op_gemm_bench(M, N, K, epi, cfg=cfg, iters=iters):.2f} us")
