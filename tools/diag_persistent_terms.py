import sys, numpy as np
sys.path.insert(0, str(__import__("pathlib").Path(__file__).resolve().parents[1]))
import gstreamer_vit_tracker_amd as vt
rng = np.random.default_rng(1)
M, N, K = 300*71, 3072, 256
bits = vt.weights.f32_to_bf16_bits
a = rng.integers(-3, 4, size=(M, K)).astype(np.float32)
w = rng.integers(-3, 4, size=(N, K)).astype(np.float32)
acc = a @ w.T
def run(rs, cs, bias, tag):
    got = vt.op_gemm_bf16(bits(a), bits(w), bias, epilogue=3, cfg=19, rowstat=rs, colsum=cs)
    y = rs[:, :1]*acc + (rs[:, 1:]*cs[None,:] + bias[None,:])
    # compare pre-relu where y>0 and got>0
    ref = np.maximum(y,0)
    refb = vt.weights.bf16_bits_to_f32(bits(ref))
    bad = got != refb
    print(tag, "bad frac", bad.mean(), "bad rows frac", bad.any(1).mean(), "bad cols frac", bad.any(0).mean())
    if bad.any():
        r,c = np.argwhere(bad)[0]; print("  first bad", r, c, got[r,c], refb[r,c], "acc", acc[r,c], "rs", rs[r], "cs", cs[c], "bias", bias[c])
        # which tiles are bad
        tb = bad.reshape(M//1, N)[:(M//256)*256].reshape(M//256,256,N//256,256).any(axis=(1,3))
        print("  bad tiles (row tiles x col tiles):", tb.sum(), "of", tb.size); print(tb[:8].astype(int))
        # within first bad tile: which rows/cols
        tr, tc = np.argwhere(tb)[0]
        sub = bad[tr*256:(tr+1)*256, tc*256:(tc+1)*256]
        print("  in tile", tr, tc, "bad rows", np.where(sub.any(1))[0][:40], "bad cols", np.where(sub.any(0))[0][:40])
ones = np.stack([np.ones(M), np.zeros(M)],1).astype(np.float32)
cs0 = np.zeros(N, np.float32); b0 = np.zeros(N, np.float32)
bias = rng.integers(-8, 9, size=N).astype(np.float32)
cs = rng.integers(-5, 6, size=N).astype(np.float32)
run(ones, cs0, b0, "identity terms")
run(ones, cs0, bias, "bias only")
rs_a = np.stack([rng.integers(1,4,size=M), np.zeros(M)],1).astype(np.float32)
run(rs_a, cs0, b0, "row scale only")
rs_b = np.stack([np.ones(M), rng.integers(-2,3,size=M)],1).astype(np.float32)
run(rs_b, cs, b0, "row shift * colsum")
