"""A/B of BUILDS of the library on one GEMM shape, alternating in one process (the comparison that survives
box-to-box and launch-to-launch variance): python tools/lib_ab.py M N K EPI CFG libA.so,libB.so[,..] [rounds]"""
import ctypes
import sys
import numpy as np
import torch  # noqa: F401  (one HIP runtime per process, see gstreamer-vit-tracker_amd/__init__.py)
M, N, K, epi, cfg = (int(v) for v in sys.argv[1:6])
paths = sys.argv[6].split(",")
rounds = int(sys.argv[7]) if len(sys.argv) > 7 else 9
libs = []
for p in paths:
    L = ctypes.CDLL(p)
    L.vt_op_gemm_bench.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
    libs.append(L)
t = {p: [] for p in paths}
for r in range(rounds):
    for p, L in zip(paths, libs):
        us = ctypes.c_float()
        rc = L.vt_op_gemm_bench(0, M, N, K, epi, cfg, 20, ctypes.byref(us))
        assert rc == 0, rc
        t[p].append(us.value)
for p in paths:
    print(f"M {M} N {N} K {K} epi {epi} cfg {cfg} {p.split('/')[-1]:36s} med {np.median(t[p]):7.2f} min {min(t[p]):7.2f} us", flush=True)
