"""How much of the host->device copy time of a rocprofv3 trace ran while a kernel of the same process
was executing: python tools/trace_overlap.py <dir with *_kernel_trace.csv and *_memory_copy_trace.csv> [min_bytes]"""
import csv, glob, os, sys
root = sys.argv[1]
min_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
kf = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
cf = glob.glob(os.path.join(root, "**", "*memory_copy_trace.csv"), recursive=True)
if not kf or not cf:
    sys.exit("trace CSVs not found under " + root)
kern, named = [], []
for r in csv.DictReader(open(kf[0])):
    kern.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    named.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Kernel_Name", "")))
kern.sort()
named.sort()
gap = int(sys.argv[3]) if len(sys.argv) > 3 else 0   # ns: kernels closer than this belong to one busy span
merged = []                             # (the tracer itself puts ~10 us between consecutive dispatches)
for a, b in kern:                       # union of kernel intervals
    if merged and a <= merged[-1][1] + gap:
        merged[-1][1] = max(merged[-1][1], b)
    else:
        merged.append([a, b])
rows = list(csv.DictReader(open(cf[0])))
tot = ov = n = 0
byts = 0
for r in rows:
    d = r.get("Direction", "")
    if "HOST_TO_DEVICE" not in d.upper() and "H2D" not in d.upper():
        continue
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    size = int(float(r.get("Size", r.get("Bytes", 0)) or 0))     # ROCm 7.2's CSV has no size column
    if (size and size < min_bytes) or (not size and b - a < 20000):   # then: skip copies under 20 us
        continue
    n += 1; tot += b - a; byts += size
    for x, y in merged:
        if y <= a: continue
        if x >= b: break
        ov += min(b, y) - max(a, x)
span = merged[-1][1] - merged[0][0]
busy = sum(y - x for x, y in merged)
what = (f"{n} host->device copies >= {min_bytes} B: {byts / 1e6:.1f} MB in {tot / 1e6:.3f} ms "
        f"({byts / max(tot, 1):.1f} GB/s while copying)") if byts else \
       f"{n} host->device copies longer than 20 us (this ROCm's CSV has no size column): {tot / 1e6:.3f} ms"
print(f"{len(kern)} kernel dispatches over {span / 1e6:.2f} ms (GPU busy with kernels {busy / span * 100:.1f} % of it); "
      f"{what}; {ov / max(tot, 1) * 100:.1f} % of that copy time "
      f"overlapped kernel execution (kernels closer than {gap / 1e3:.0f} us counted as one busy span)")


# Pass view: a pass of the engine runs from its crop/resize kernel (preproc over all streams: > 30 us)
# to its decode kernel. Where does each upload sit relative to the passes, and how long does the GPU
# wait between one pass's last kernel and the next pass's first?
passes, cur = [], None
for a, b, nm in named:
    if "preproc" in nm and b - a > 30000:
        cur = [a, None]
    elif "decode_kernel" in nm and cur is not None:
        cur[1] = b
        passes.append(tuple(cur))
        cur = None
if len(passes) > 3:
    inside = before = 0
    for r in rows:
        d = r.get("Direction", "").upper()
        a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if "HOST_TO_DEVICE" not in d or b - a < 20000:
            continue
        if any(x <= a and b <= y for x, y in passes):
            inside += 1
        else:
            before += 1
    gaps = [passes[i + 1][0] - passes[i][1] for i in range(2, len(passes) - 1)]
    dur = [y - x for x, y in passes[2:]]
    print(f"{len(passes)} passes of {sum(dur) / len(dur) / 1e6:.2f} ms; uploads lying entirely INSIDE a running pass: {inside}, "
          f"outside one (includes the uploads of the init calls and of the first pass): {before}; idle time between consecutive passes: median "
          f"{sorted(gaps)[len(gaps) // 2] / 1e3:.0f} us, max {max(gaps) / 1e3:.0f} us")
