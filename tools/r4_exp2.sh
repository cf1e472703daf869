#!/bin/bash
# round-4 experiment 2: write-through (sc1) output stores in the persistent GEMM, with and without XCD ownership
set -u
# builds first: python build.py --variant base -DVT_AB_PLAINOUT; --variant own -DVT_AB_OWN -DVT_AB_PLAINOUT; --variant sc1 (today's default);
#               --variant ownsc1 -DVT_AB_OWN   (when this script ran, write-through stores were the flag -DVT_AB_SC1OUT and plain stores the default)
P=gstreamer-vit-tracker_amd
OUT=gpurun_out/r4_exp2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
LIBS=$P/libvittrack_hip.so,$P/libvittrack_hip_own.so,$P/libvittrack_hip_sc1.so,$P/libvittrack_hip_ownsc1.so
python3 tools/lib_ab.py 21600 3072 768 2 19 $LIBS 9 2>&1 | grep -v amdgpu.ids | tee $OUT/fc1_ab.txt
python3 tools/lib_ab.py 21600 2304 768 4 19 $LIBS 9 2>&1 | grep -v amdgpu.ids | tee $OUT/qkv_ab.txt
python3 tools/lib_ab.py 32340 4096 1024 2 19 $LIBS 5 2>&1 | grep -v amdgpu.ids | tee $OUT/fc1_cfg5_ab.txt
for v in _sc1 _ownsc1; do
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
