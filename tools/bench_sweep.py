"""bench.py over (streams per GPU, engines) combinations, one line each:
   python tools/bench_sweep.py 60x2,30x1,31x1,31xauto,45x1,45xauto,62x2,90x2,90x3 [steps]
   ('auto': engines as vt_plan_engines splits them)"""
import json, subprocess, sys
combos = sys.argv[1].split(",") if len(sys.argv) > 1 else ["60x2", "30x1"]
steps = sys.argv[2] if len(sys.argv) > 2 else "60"
for c in combos:
    s, g = c.split("x")
    eng = ["--engines", "auto"] if g == "auto" else ["--groups", g]
    r = subprocess.run([sys.executable, "bench.py", "--steps", steps, "--warmup", "10", "--streams", s, *eng,
                        "--no-cpu-baseline", "--no-profile", "--no-host-leg"], capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        per_frame_us = d["ms_per_step"] * 1e3 / int(s)
        c = c + " = " + "+".join(str(x) for x in d["config"]["engine_sizes"])
        print(f"{c:>16}: {d['value']:8.1f} frames/s  {d['ms_per_step']:7.3f} ms/step  {per_frame_us:6.2f} us/frame  "
              f"whole-frame MFMA {d['whole_frame_mfma_frac']*100:5.1f} %  tracked_ok {d['tracked_ok']}", flush=True)
    except Exception as e:
        print(c, "failed:", e, r.stderr[-500:], flush=True)
