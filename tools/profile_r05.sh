#!/bin/bash
# round-5 measurement set on one MI355X box (run from the repository root): rocprofv3 kernel-trace summaries of
# bench.py (one engine / two engines), PMC passes of the dominant kernel, the attention kernel and the head's band
# kernel, the bench lines (default, one engine, cfg2, cfg5, the literal cfg4 shape at N = 1). Results under
# gpurun_out/r05/; the summaries that are judged are copied into profiles/ afterwards.
#   PART=1 bash tools/profile_r05.sh   (traces + PMC)      PART=2 bash tools/profile_r05.sh   (bench lines)
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r05
mkdir -p $O
if [ "${PART:-1}" = "1" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_30x1 -- python3 bench.py --steps 60 --warmup 10 --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_30x1.json 2> $O/trace_30x1.err
echo "trace 30x1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_60x2 -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_60x2.json 2> $O/trace_60x2.err
echo "trace 60x2 done"
OUT=gpurun_out/pmc_r05
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
M=21600; MS=17280
run fc1 python3 tools/one_gemm.py $M 3072 768 2 19 20
run attn python3 tools/attn_bench.py 3 30
run conv3 python3 tools/one_headconv.py 30 24 128 128 1 0 0 20
run conv1 python3 tools/one_headconv.py 30 24 768 128 0 0 0 20
run lnconv1 python3 tools/one_headconv.py 30 24 768 128 2 0 0 20
python3 tools/pmc_summary.py $OUT/fc1 gemm256p_kernel $O/r05_dominant_kernel_pmc.json --family gemm_bf16_gelu_256x256pp_n3072k768 \
    --streams 30 --algorithmic-bytes $((M*768*2 + 3072*768*2 + M*3072*2)) --command "python3 tools/one_gemm.py $M 3072 768 2 19 20" > /dev/null
python3 tools/pmc_summary.py $OUT/attn attention_dma_kernel $O/r05_attention_pmc.json --family attention \
    --streams 30 --algorithmic-bytes $((M*768*2*4)) --command "python3 tools/attn_bench.py 3 30" > /dev/null
python3 tools/pmc_summary.py $OUT/conv3 head_conv_kernel $O/r05_head_conv3x3_pmc.json --family head_conv3x3 \
    --streams 30 --algorithmic-bytes $((MS*128*2*2 + 128*1152*2)) --command "python3 tools/one_headconv.py 30 24 128 128 1 0 0 20" > /dev/null
python3 tools/pmc_summary.py $OUT/conv1 head_conv_kernel $O/r05_head_conv1x1_pmc.json --family head_conv1x1 \
    --streams 30 --algorithmic-bytes $((MS*768*2 + MS*128*2 + 128*768*2)) --command "python3 tools/one_headconv.py 30 24 768 128 0 0 0 20" > /dev/null
python3 tools/pmc_summary.py $OUT/lnconv1 head_conv_kernel $O/r05_head_ln_conv1x1_pmc.json --family head_ln_conv1x1 \
    --streams 30 --algorithmic-bytes $((MS*768*4 + MS*128*2 + 128*768*2)) --command "python3 tools/one_headconv.py 30 24 768 128 2 0 0 20" > /dev/null
echo "pmc done"
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
