"""config 16 (128x192 tiles, two workgroups per CU) against config 2 on the same operands, bit for bit
(bf16 epilogues have the same summation order per output element: one MFMA chain over K), and timings.
    python3 tools/g128_check.py"""
import sys
import numpy as np
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))
import gstreamer_vit_tracker_amd as vt
from gstreamer_vit_tracker_amd.weights import f32_to_bf16_bits

rng = np.random.default_rng(5)
for (M, N, K, epi, ln) in [(757, 768, 384, 2, False), (757, 768, 384, 3, False), (1000, 576, 768, 2, True), (128, 192, 64, 3, False),
                           (130, 960, 3072, 2, True)]:
    a = f32_to_bf16_bits(rng.standard_normal((M, K)).astype(np.float32))
    w = f32_to_bf16_bits((rng.standard_normal((N, K)) * 0.05).astype(np.float32))
    bias = rng.standard_normal(N).astype(np.float32)
    kw = {}
    if ln:
        kw = dict(rowstat=np.stack([rng.uniform(0.5, 2, M), rng.standard_normal(M)], 1).astype(np.float32),
                  colsum=rng.standard_normal(N).astype(np.float32))
    ref = vt.op_gemm_bf16(a, w, bias, epilogue=epi, cfg=2, **kw)
    got = vt.op_gemm_bf16(a, w, bias, epilogue=epi, cfg=16, **kw)
    print(f"M {M} N {N} K {K} epi {epi} ln {ln}: identical {np.array_equal(ref, got)}  max |d| {np.abs(ref - got).max():.3g}", flush=True)
for (M, N, K, epi, name) in [(21600, 3072, 768, 2, "fc1 cfg3 x30"), (43200, 3072, 768, 2, "fc1 cfg3 x60"), (29400, 3072, 1024, 2, "cfg5-like x30")]:
    for cfg in (16, 19):
        us = vt.op_gemm_bench(M, N, K, epi, cfg=cfg, iters=30)
        print(f"{name}: cfg {cfg}: {us:.1f} us  {2.0 * M * N * K / us * 1e-6:.0f} TF", flush=True)
