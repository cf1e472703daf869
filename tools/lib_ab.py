"""A/B of two BUILDS of the library on one GEMM shape, both loaded into ONE process and timed in
interleaved rounds:  python tools/lib_ab.py libA.so libB.so [streams] [epilogue] [cfg] [rounds]
(epilogue: library numbering, 2 = GELU; N and K follow from it for ViT-B: 2 -> fc1, 4 -> QKV)."""
import ctypes, sys
import numpy as np
a, b = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 30
epi = int(sys.argv[4]) if len(sys.argv) > 4 else 2
cfg = int(sys.argv[5]) if len(sys.argv) > 5 else 19
rounds = int(sys.argv[6]) if len(sys.argv) > 6 else 7
M, D = 720 * B, 768
N, K = {2: (4 * D, D), 3: (4 * D, D), 4: (3 * D, D), 1: (D, 4 * D)}[epi]
libs = {}
for p in (a, b):
    L = ctypes.CDLL(p)
    L.vt_op_gemm_bench.argtypes = [ctypes.c_int] * 7 + [ctypes.POINTER(ctypes.c_float)]
    libs[p] = L
t = {p: [] for p in libs}
for r in range(rounds):
    for p, L in libs.items():
        us = ctypes.c_float()
        rc = L.vt_op_gemm_bench(0, M, N, K, epi, cfg, 20, ctypes.byref(us))
        assert rc == 0, rc
        t[p].append(us.value)
for p in libs:
    print(f"{p:>50}: med {np.median(t[p]):7.1f} min {min(t[p]):7.1f} us  {2.0 * M * N * K / np.median(t[p]) / 1e6:5.0f} TF "
          f"(M={M} N={N} K={K} epi {epi} cfg {cfg})", flush=True)
