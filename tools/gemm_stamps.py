"""In-kernel cycle stamps of the GEMM kernels (diagnostic build): 
   python gstreamer-vit-tracker_amd/build.py --stamps && VITTRACK_HIP_LIB=.../libvittrack_hip_stamps.so python tools/gemm_stamps.py 19 30
prints (stderr of the library) per-wave mean cycles: wait / epilogue(issue) / main(compute) / total."""
import os, sys
sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
os.environ.setdefault("VITTRACK_HIP_LIB", os.path.join("gstreamer-vit-tracker_amd", "libvittrack_hip_stamps.so"))
import gstreamer_vit_tracker_amd as vt
cfgs = [int(c) for c in sys.argv[1].split(",")] if len(sys.argv) > 1 else [19]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 30
M = 720 * B
for (N, K, epi, name) in [(2304, 768, 4, "qkv"), (3072, 768, 2, "gelu"), (768, 3072, 1, "fc2"), (768, 768, 1, "proj")]:
    for c in cfgs:
        us = vt.op_gemm_bench(M, N, K, epi, c, iters=10)
        print(f"{name} cfg {c}: {us:.1f} us", flush=True)
