"""print the headline numbers and the per-kernel table of bench.py JSON lines: python tools/show_bench.py FILE..."""
import json
import sys

for f in sys.argv[1:]:
    lines = [l for l in open(f).read().strip().splitlines() if l.startswith("{")]
    if not lines:
        print(f, "no JSON line")
        continue
    d = json.loads(lines[-1])
    r = d.get("roofline", {})
    print(f"{f}: {d['value']:.1f} frames/s, {d['ms_per_step']:.3f} ms/step, whole-frame MFMA {d['whole_frame_mfma_frac']:.4f}, "
          f"dominant {r.get('kernel')} {r.get('frac', 0):.4f} ({r.get('avg_launch_us', 0):.1f} us), sync latency "
          f"{d['sync_update_latency_ms']:.3f} ms, tracked_ok {d['tracked_ok']}, eager sum {d.get('eager_event_ms_per_step', 0):.3f} ms")
    for k in d.get("kernels", []):
        print(f"   {k['name']:<46} x{k['launches']:<3} {k['ms']:8.4f} ms {100 * k['share']:5.1f}%  {k['tflops']:7.1f} TF")
    for key in ("single_stream", "pcie_inclusive", "cpu_baseline"):
        if key in d:
            print("  ", key, json.dumps(d[key])[:400])
