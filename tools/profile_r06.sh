#!/bin/bash
# round-6 measurement set on one MI355X box (run from the repository root): rocprofv3 kernel-trace summaries of bench.py
# (one engine / two engines), PMC passes of the dominant kernel (tied to the kernel build through vt_build_info), the
# per-kernel power table, the bench lines. Results under gpurun_out/r06/; what is judged is copied into profiles/ afterwards.
#   PART=1 bash tools/profile_r06.sh   (traces + PMC)      PART=2 ...   (bench lines)      PART=3 ...   (the driver's command x 3)      PART=4 ...   (PMC alone)
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r06
mkdir -p $O
if [ "${PART:-1}" = "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_30x1 -- python3 bench.py --steps 60 --warmup 10 --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_30x1.json 2> $O/trace_30x1.err
echo "trace 30x1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_60x2 -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_60x2.json 2> $O/trace_60x2.err
echo "trace 60x2 done"
fi
if [ "${PART:-1}" = "1" ] || [ "${PART:-1}" = "4" ]; then     # PMC passes of the dominant kernel (PART=4: these alone)
OUT=gpurun_out/pmc_r06
CGROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"
        "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE")
run() {   # tag, command...
    local tag=$1; shift
    local i=0
    for c in "${CGROUPS[@]}"; do
        rocprofv3 --pmc $c --output-format csv -d $OUT/$tag/g$i -- "$@" > $OUT/$tag.g$i.log 2>&1
        i=$((i + 1))
    done
}
mkdir -p $OUT
M=21600
SHA=$(python3 -c "import gstreamer_vit_tracker_amd as vt; print(vt.build_info()['k_gemm256'])" 2>/dev/null | tail -1)
run fc1 python3 tools/one_gemm.py $M 3072 768 2 19 20
python3 tools/pmc_summary.py $OUT/fc1 gemm256p_kernel $O/r06_dominant_kernel_pmc.json --family gemm_bf16_gelu_256x256pp_n3072k768 \
    --streams 30 --algorithmic-bytes $((M*768*2 + 3072*768*2 + M*3072*2)) --command "python3 tools/one_gemm.py $M 3072 768 2 19 20" --kernel-sha "$SHA" > /dev/null
echo "pmc done ($SHA)"
find $O -name "*kernel_stats.csv" | head
fi
if [ "${PART:-1}" = "2" ]; then
python3 bench.py > $O/bench_cfg3_60x2.json 2> $O/bench_cfg3_60x2.err
echo "bench default done"
python3 bench.py --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/bench_cfg3_30x1.json 2>> $O/bench.err
python3 bench.py --workload cfg2 --no-cpu-baseline --no-single-leg > $O/bench_cfg2.json 2>> $O/bench.err
python3 bench.py --workload cfg5 --steps 100 --no-cpu-baseline --no-single-leg > $O/bench_cfg5.json 2>> $O/bench.err
python3 bench.py --streams 1 --groups 1 --steps 1000 --warmup 100 --no-cpu-baseline --no-host-leg --no-single-leg > $O/bench_cfg4_literal_n1.json 2>> $O/bench.err
echo "benches done"
fi
if [ "${PART:-1}" = "3" ]; then     # the driver's exact command, three times
for i in 1 2 3; do python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_cmd_$i.json 2>> $O/driver_cmd.err; done
python3 - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r06/driver_cmd_*.json")):
    d = json.loads(open(f).read().strip().split("\n")[-1])
    print(f, round(d["value"], 1), round(d["whole_frame_mfma_frac"], 4), round(d["roofline"]["frac"], 4), d["roofline"]["traffic"])
PY
fi
