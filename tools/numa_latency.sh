#!/bin/bash
# one tracker's synchronous update latency with the process pinned to the GPU's NUMA-local CPUs, to the other node's, or not at all
# (run on the GPU box from the repository root)
LOCAL=$(python3 - <<'PY'
import importlib.util, os
spec = importlib.util.spec_from_file_location("pl", "gstreamer-vit-tracker_amd/placement.py"); pl = importlib.util.module_from_spec(spec); spec.loader.exec_module(pl)
lists, vis = pl.gpu_local_cpulists(), pl.visible_device_map()
allowed = os.sched_getaffinity(0)
dev = vis[0] if vis else 0
loc = set(lists[dev]) if dev < len(lists) else set()
import sys
print("KFD GPU nodes:", len(lists), "visible map:", vis, "local lists:", [pl.format_cpulist(l) for l in lists], file=sys.stderr)
print(pl.format_cpulist(sorted(loc & allowed)) or "", pl.format_cpulist(sorted(allowed - loc)) or "", sep=" ")
PY
)
NEAR=$(echo $LOCAL | cut -d' ' -f1); FAR=$(echo $LOCAL | cut -d' ' -f2)
echo "allowed: $(python3 -c 'import os; print(len(os.sched_getaffinity(0)))') CPUs; GPU-local: $NEAR ; others: $FAR"
numactl -H 2>/dev/null | head -12
for r in 1 2 3; do
  for where in host device; do
    echo "unpinned $where: $(python3 tools/one_tracker.py 500 $where 2>/dev/null | tail -1)"
    [ -n "$NEAR" ] && echo "near     $where: $(taskset -c $NEAR python3 tools/one_tracker.py 500 $where 2>/dev/null | tail -1)"
    [ -n "$FAR" ] && echo "far      $where: $(taskset -c $FAR python3 tools/one_tracker.py 500 $where 2>/dev/null | tail -1)"
  done
done
