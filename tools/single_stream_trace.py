"""Where one synchronous update of ONE tracker spends its time, from a rocprofv3 kernel trace of tools/one_tracker.py:
   rocprofv3 --kernel-trace --output-format csv -d gpurun_out/r06/trace_one -- python3 tools/one_tracker.py 200 device
   python tools/single_stream_trace.py gpurun_out/r06/trace_one
Per update (the dispatches between two crop kernels): span first start -> last end, sum of the kernel durations, and the
boundaries between consecutive dependent kernels (end -> next start); per kernel family mean duration and mean boundary
in front of it. The tracer adds its own cost to every dispatch: the boundary numbers are an UPPER bound of the untraced ones."""
import csv
import glob
import os
import sys
import collections
import statistics

root = sys.argv[1]
f = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
if not f:
    sys.exit("no kernel trace under " + root)
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f[0]))))
# an update starts at every search-window crop kernel (template crops at init are dropped with the first updates)
starts = [i for i, r in enumerate(rows) if "preproc" in r[2]]
updates = [rows[a:b] for a, b in zip(starts, starts[1:])]
n_k = statistics.mode(len(u) for u in updates)
updates = [u for u in updates if len(u) == n_k][20:]          # steady state, complete updates only
short = lambda name: name.split("(")[0].split("<")[0].replace("void ", "")[:40]
span = [u[-1][1] - u[0][0] for u in updates]
busy = [sum(e - s for s, e, _ in u) for u in updates]
gaps = [sum(max(0, u[i + 1][0] - u[i][1]) for i in range(len(u) - 1)) for u in updates]
period = [b[0][0] - a[0][0] for a, b in zip(updates, updates[1:])]
us = lambda v: statistics.median(v) / 1e3
print(f"{len(updates)} steady-state updates of {n_k} dispatches each (medians, microseconds):")
print(f"  first kernel start -> last kernel end   {us(span):8.1f}")
print(f"  sum of kernel durations                 {us(busy):8.1f}")
print(f"  sum of boundaries (end -> next start)   {us(gaps):8.1f}   = {us(gaps) / (n_k - 1):.2f} per boundary")
print(f"  update period (crop start -> next crop) {us(period):8.1f}   (host side between updates: {us(period) - us(span):.1f})")
fam = collections.OrderedDict()
for u in updates:
    for i, (s, e, name) in enumerate(u):
        d = fam.setdefault(short(name), [0, 0, 0])
        d[0] += 1; d[1] += e - s
        if i:
            d[2] += max(0, s - u[i - 1][1])
print("  per kernel family: launches per update, mean duration, mean boundary in front")
for k, (n, dur, gap) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
    print(f"    {k:42s} {n / len(updates):5.1f} x {dur / n / 1e3:7.2f} us   + {gap / n / 1e3:5.2f} us")
