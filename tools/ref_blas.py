"""Reference point only (not used by the product): torch.matmul bf16 (hipBLASLt / rocBLAS) on the
tracker's GEMM shapes, random operands, plain C = A W^T without any of the fused epilogues."""
import sys, torch
M = int(sys.argv[1]) if len(sys.argv) > 1 else 21600
torch.manual_seed(0)
for (N, K, name) in [(2304, 768, "qkv"), (768, 768, "proj"), (3072, 768, "fc1"), (768, 3072, "fc2")]:
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        c = a @ w.t()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        c = a @ w.t()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print(f"{name:5s} M={M} N={N} K={K}: {us:7.1f} us  {2.0 * M * N * K / us / 1e6:6.0f} TF  (library GEMM, bf16 out, no epilogue)")
