import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
B, mode, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
us = vt.op_attention_bench(B, 720, 12, mode, iters=iters)
print(f"B={B} mode={mode}: {us:.1f} us")
