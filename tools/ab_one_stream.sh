#!/bin/bash
# same-box A/B of two library builds on the one-stream legs of bench.py (alternating). The build to compare against is expected as
# gstreamer-vit-tracker_amd/libvittrack_hip_before.so: check the other revision out into a git worktree, run __graft_entry__.build() there and
# copy its libvittrack_hip.so under that name (built .so files travel to the GPU box with the tree)
O=gpurun_out/r06; mkdir -p $O
for r in 1 2 3; do
  for lib in libvittrack_hip_before.so libvittrack_hip.so; do
    VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/$lib python3 bench.py --streams 1 --groups 1 --steps 1000 --warmup 100 --no-cpu-baseline --no-host-leg > $O/ab_one_$lib.$r.json 2>>$O/ab_one.err
    python3 - "$O/ab_one_$lib.$r.json" "$lib" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().split("\n")[-1]); s = d["single_stream"]
print(f"{sys.argv[2]:32s} value {d['value']:7.1f} updates/s ({d['ms_per_step']:.4f} ms)   sync p50: host {s['host_pointer']['ms_p50']:.4f}  registered {s['host_pointer_registered']['ms_p50']:.4f}  device {s['device_pointer']['ms_p50']:.4f} ms", flush=True)
PY
  done
done
