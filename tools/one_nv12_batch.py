"""the batched converter on device-resident random frames, for profilers: python tools/one_nv12_batch.py W H N ITERS"""
import sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt
w, h, n, iters = (int(v) for v in sys.argv[1:5])
print(f"{vt.op_nv12_to_rgb8_batch_bench(w, h, n, iters=iters):.2f} us")
