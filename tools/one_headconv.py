"""one launch shape of the head's band kernel on device-resident pseudo-random operands, for profilers and the stamps
build: python tools/one_headconv.py B GRID CIN N CONV3X3 R NCB ITERS   (R / NCB 0: the launcher's plan; CONV3X3 = 2: the 1x1
layer with the final LayerNorm inside, CIN = D, the rows of a stream at the offset of a (GRID/2)^2-token template)"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
B, grid, cin, n, conv, R, ncb, iters = (int(v) for v in sys.argv[1:9])
if conv == 2:
    off = (grid // 2) ** 2
    print(f"{vt.op_headconv_ln_bench(B, grid, cin, n, grid * grid + off, off, fused=True, R=R, ncb=ncb, iters=iters):.2f} us")
else:
    print(f"{vt.op_headconv_bench(B, grid, cin, n, bool(conv), R, ncb, iters=iters):.2f} us")
