#!/bin/bash
# round-4 experiments, one GPU session: (1) fc1 tile order / A-operand cache policy (time, alternating builds in one
# process, then fabric reads and L2 hit rate per build), (2) what cutting the residual pair's bytes buys the X-epilogues
# builds first (in the container; the .so files travel with the snapshot):
#   cd gstreamer-vit-tracker_amd && for v in "own -DVT_AB_OWN" "ownnt -DVT_AB_OWN -DVT_AB_ANT" "ant -DVT_AB_ANT" "nolo -DVT_AB_NOLO" "lo8 -DVT_AB_LO8"; do
#       set -- $v; n=$1; shift; python build.py --variant $n "$@"; done
# (round 4's base build for this script had plain output stores: add -DVT_AB_PLAINOUT to reproduce its "base")
set -u
P=gstreamer-vit-tracker_amd
OUT=gpurun_out/r4_exp1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
python3 tools/lib_ab.py 21600 3072 768 2 19 $P/libvittrack_hip.so,$P/libvittrack_hip_own.so,$P/libvittrack_hip_ownnt.so,$P/libvittrack_hip_ant.so 9 2>&1 | grep -v amdgpu.ids | tee $OUT/fc1_ab.txt
python3 tools/lib_ab.py 21600 2304 768 4 19 $P/libvittrack_hip.so,$P/libvittrack_hip_own.so,$P/libvittrack_hip_ownnt.so,$P/libvittrack_hip_ant.so 9 2>&1 | grep -v amdgpu.ids | tee $OUT/qkv_ab.txt
for shape in "21600 768 768" "21600 768 3072"; do
    python3 tools/lib_ab.py $shape 1 18 $P/libvittrack_hip.so,$P/libvittrack_hip_nolo.so,$P/libvittrack_hip_lo8.so 9 2>&1 | grep -v amdgpu.ids | tee -a $OUT/xepi_ab.txt
done
for v in "" _own _ownnt _ant; do
    export VITTRACK_HIP_LIB=$PWD/$P/libvittrack_hip$v.so
    i=0
    for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE"; do
        rocprofv3 --pmc $c --output-format csv -d $OUT/pmc_fc1$v/g$i -- python3 tools/one_gemm.py 21600 3072 768 2 19 20 > $OUT/pmc_fc1$v.g$i.log 2>&1
        i=$((i + 1))
    done
    echo "== fc1 build '$v'" | tee -a $OUT/pmc.txt
    python3 - "$OUT/pmc_fc1$v" gemm256p <<'PY' | tee -a $OUT/pmc.txt
import csv, glob, sys, collections
out, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
a = {k: sum(v) / len(v) for k, v in acc.items()}
for k in sorted(a): print(f"{k:16s} {a[k]:14.1f} (n={len(acc[k])})")
if "FETCH_SIZE" in a: print(f"fabric reads {2*a['FETCH_SIZE']*1024/1e6:.1f} MB (algorithmic operands 37.9 MB)")
if "WRITE_SIZE" in a: print(f"writes {a['WRITE_SIZE']*1024/1e6:.1f} MB")
if "TCC_HIT_sum" in a: print(f"L2 hit {a['TCC_HIT_sum']/(a['TCC_HIT_sum']+a['TCC_MISS_sum']):.3f}")
PY
done
unset VITTRACK_HIP_LIB
