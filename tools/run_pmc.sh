#!/bin/bash
# PMC passes of one command (each counter group in its own rocprofv3 run): tools/run_pmc.sh <outdir> <cmd...>
out=$1; shift
mkdir -p "$out"
i=0
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE" \
         "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
         "SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $c --output-format csv -d "$out/p$i" -- "$@" > "$out/p$i.log" 2>&1 || { echo "pass $i ($c) failed"; tail -5 "$out/p$i.log"; exit 1; }
  echo "pass $i ($c) ok"
done
