#!/bin/bash
# (round 4: the same passes on the round-4 kernels) rocprofv3 --pmc passes of the encoder kernels at 30 cfg3 streams, one counter group per run (gpurun refuses
# --pmc together with the trace domains); summaries -> gpurun_out/r04/r04_*_pmc.json (tools/pmc_summary.py).
#   bash tools/pmc_r04.sh            (on the GPU box, from the repository root)
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
OUT=gpurun_out/pmc_r04
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
mkdir -p $OUT gpurun_out/r04
M=21600
run fc1 python3 tools/one_gemm.py $M 3072 768 2 19 20
run qkv python3 tools/one_gemm.py $M 2304 768 4 19 20
run proj python3 tools/one_gemm.py $M 768 768 1 18 20
run fc2 python3 tools/one_gemm.py $M 768 3072 1 18 20
run attn python3 tools/attn_bench.py 3 30
# algorithmic bytes: operands once + outputs once (bf16 pair of the residual stream: 4 B read + 4 B written)
python3 tools/pmc_summary.py $OUT/fc1 gemm256p_kernel gpurun_out/r04/r04_dominant_kernel_pmc.json --family gemm_bf16_gelu_256x256pp_n3072k768 \
    --streams 30 --algorithmic-bytes $((M*768*2 + 3072*768*2 + M*3072*2)) --command "python3 tools/one_gemm.py $M 3072 768 2 19 20"
python3 tools/pmc_summary.py $OUT/qkv gemm256p_kernel gpurun_out/r04/r04_qkv_pmc.json --family gemm_bf16_qkv_256x256pp_n2304k768 \
    --streams 30 --algorithmic-bytes $((M*768*2 + 2304*768*2 + M*2304*2)) --command "python3 tools/one_gemm.py $M 2304 768 4 19 20"
python3 tools/pmc_summary.py $OUT/proj gemm256_kernel gpurun_out/r04/r04_proj_pmc.json --family gemm_bf16_xresid_256x256pp_n768k768 \
    --streams 30 --algorithmic-bytes $((M*768*2 + 768*768*2 + M*768*8)) --command "python3 tools/one_gemm.py $M 768 768 1 18 20"
python3 tools/pmc_summary.py $OUT/fc2 gemm256_kernel gpurun_out/r04/r04_fc2_pmc.json --family gemm_bf16_xresid_256x256pp_n768k3072 \
    --streams 30 --algorithmic-bytes $((M*3072*2 + 768*3072*2 + M*768*8)) --command "python3 tools/one_gemm.py $M 768 3072 1 18 20"
python3 tools/pmc_summary.py $OUT/attn attention_dma_kernel gpurun_out/r04/r04_attention_pmc.json --family attention \
    --streams 30 --algorithmic-bytes $((M*768*2*4)) --command "python3 tools/attn_bench.py 3 30"
