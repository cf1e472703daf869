#!/bin/bash
# rocprofv3 --pmc passes (one counter group per run) of one GEMM: bash tools/pmc_one.sh TAG M N K EPI CFG KERNEL_SUBSTR
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
TAG=$1; shift
OUT=gpurun_out/pmc_one/$TAG
mkdir -p $OUT
CG=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
    "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU"
    "GRBM_GUI_ACTIVE" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum")
i=0
for c in "${CG[@]}"; do
    rocprofv3 --pmc $c --output-format csv -d $OUT/g$i -- python3 tools/one_gemm.py $1 $2 $3 $4 $5 20 > $OUT/g$i.log 2>&1
    i=$((i + 1))
done
python3 - "$OUT" "$6" <<'PY'
import csv, glob, sys, collections
out, sub = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/g*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    v = acc[k]
    print(f"{k:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
PY
