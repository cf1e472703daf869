"""one GEMM shape / config for rocprofv3 (python tools/one_gemm.py M N K epi cfg iters)"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
M, N, K, epi, cfg, iters = (int(x) for x in sys.argv[1:7])
us = vt.op_gemm_bench(M, N, K, epi, cfg, iters=iters)
print(f"M={M} N={N} K={K} epi={epi} cfg={cfg}: {us:.1f} us  {2.0*M*N*K/us/1e6:.0f} TF")
