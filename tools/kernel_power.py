"""Socket power and shader clock while ONE kernel of the pass loops alone on the chip (rocm-smi sampled beside an
operator-level timing loop): which kernels run AT the 1,400-W package limit (their time is energy), which below it
(their time is structure), and the energy of one launch = power x time.
   python tools/kernel_power.py [seconds per kernel] [name,name]   (on the GPU box; VITTRACK_HIP_LIB selects a tuning build)"""
import re
import subprocess
import sys
import threading
import time

sys.path.insert(0, str(__import__('pathlib').Path(__file__).resolve().parents[1]))   # repo root
import gstreamer_vit_tracker_amd as vt

SECS = float(sys.argv[1]) if len(sys.argv) > 1 else 5.0
ONLY = sys.argv[2].split(",") if len(sys.argv) > 2 else None       # substrings of the kernel names to run
B = 30
M = 720 * B


def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=10).stdout
            pw = re.search(r"Power[^\n]*?:\s*([\d.]+)\s*$", txt, re.M)
            sc = re.search(r"sclk[^\n]*?\((\d+)Mhz\)", txt)
            if pw and sc:
                out.append((float(pw.group(1)), int(sc.group(1))))
        except Exception:
            pass
        time.sleep(0.25)


def run(name, flops, byts, fn_us):
    us0 = fn_us(20)                          # warm-up + a first estimate
    iters = max(50, int(SECS * 1e6 / us0))
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    us = fn_us(iters)
    stop.set()
    th.join()
    mid = out[len(out) // 4:] or out          # drop the ramp
    pw = sum(p for p, _ in mid) / max(len(mid), 1)
    sc = sum(s for _, s in mid) / max(len(mid), 1)
    print(f"{name:34s} {us:8.1f} us  {flops / us / 1e6:7.0f} TFLOP/s  {byts / us / 1e6:6.2f} TB/s alg.  "
          f"{pw:6.0f} W  sclk {sc:5.0f} MHz  energy/launch {pw * us * 1e-3:7.1f} mJ  ({len(mid)} samples)", flush=True)


D, MLP = 768, 3072
kernels = [
    ("fc1 + GELU (256x256 persistent)", 2.0 * M * MLP * D, 2.0 * (M * D + MLP * D + M * MLP), lambda it: vt.op_gemm_bench(M, MLP, D, 2, -1, iters=it)),
    ("QKV (256x256 persistent)", 2.0 * M * 3 * D * D, 2.0 * (M * D + 3 * D * D + M * 3 * D), lambda it: vt.op_gemm_bench(M, 3 * D, D, 4, -1, iters=it)),
    ("fc2 + residual (256x256)", 2.0 * M * D * MLP, 2.0 * (M * MLP + D * MLP) + 6.0 * M * D, lambda it: vt.op_gemm_bench(M, D, MLP, 1, -1, iters=it)),
    ("proj + residual (256x256)", 2.0 * M * D * D, 2.0 * (M * D + D * D) + 6.0 * M * D, lambda it: vt.op_gemm_bench(M, D, D, 1, -1, iters=it)),
    ("attention (mode 3)", 4.0 * B * 720 * 720 * D, 8.0 * M * D, lambda it: vt.op_attention_bench(B, 720, 12, -1, iters=it)),
    ("nv12 -> rgb8, 30 x 1080p", 0.0, 1920 * 1080 * 4.5 * 30, lambda it: vt.op_nv12_to_rgb8_batch_bench(1920, 1080, 30, iters=it)),
]
print(f"# one kernel looping alone, {SECS:.0f} s each, {B} streams (M = {M}); rocm-smi every 0.25 s, first quarter of the samples dropped")
for k in kernels:
    if ONLY is None or any(o in k[0] for o in ONLY):
        run(*k)
