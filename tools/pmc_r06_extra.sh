#!/bin/bash
# PMC passes (one counter group per run) of the other encoder kernels and of the batched converter on the round-6 tree
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r06; OUT=gpurun_out/pmc_r06x
mkdir -p $O $OUT
CGROUPS=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE")
run() { local tag=$1; shift; local i=0; for c in "${CGROUPS[@]}"; do rocprofv3 --pmc $c --output-format csv -d $OUT/$tag/g$i -- "$@" > $OUT/$tag.g$i.log 2>&1; i=$((i + 1)); done; }
M=21600
SHA=$(python3 -c "import gstreamer_vit_tracker_amd as vt; print(vt.build_info()['k_gemm256'])" 2>/dev/null | tail -1)
run qkv python3 tools/one_gemm.py $M 2304 768 4 19 20
run proj python3 tools/one_gemm.py $M 768 768 1 18 20
run fc2 python3 tools/one_gemm.py $M 768 3072 1 18 20
run attn python3 tools/attn_bench.py 3 30
run nv12b python3 tools/one_nv12_batch.py 1920 1080 30 20
python3 tools/pmc_summary.py $OUT/qkv gemm256p_kernel $O/r06_qkv_pmc.json --family gemm_bf16_qkv_256x256pp_n2304k768 --streams 30 --algorithmic-bytes $((M*768*2 + 2304*768*2 + M*2304*2)) --command "python3 tools/one_gemm.py $M 2304 768 4 19 20" --kernel-sha "$SHA" > /dev/null
python3 tools/pmc_summary.py $OUT/proj gemm256_kernel $O/r06_proj_pmc.json --family gemm_bf16_xresid_256x256pp_n768k768 --streams 30 --algorithmic-bytes $((M*768*2 + 768*768*2 + M*768*6)) --command "python3 tools/one_gemm.py $M 768 768 1 18 20" --kernel-sha "$SHA" > /dev/null
python3 tools/pmc_summary.py $OUT/fc2 gemm256_kernel $O/r06_fc2_pmc.json --family gemm_bf16_xresid_256x256pp_n768k3072 --streams 30 --algorithmic-bytes $((M*3072*2 + 768*3072*2 + M*768*6)) --command "python3 tools/one_gemm.py $M 768 3072 1 18 20" --kernel-sha "$SHA" > /dev/null
python3 tools/pmc_summary.py $OUT/attn attention_dma_kernel $O/r06_attention_pmc.json --family attention --streams 30 --algorithmic-bytes $((M*768*2*4)) --command "python3 tools/attn_bench.py 3 30" > /dev/null
python3 tools/pmc_summary.py $OUT/nv12b nv12_to_rgb8_batch_kernel $O/r06_nv12_batch_pmc.json --family nv12_to_rgb8_batch --streams 30 --algorithmic-bytes $((1920*1080*9/2*30)) --command "python3 tools/one_nv12_batch.py 1920 1080 30 20" > /dev/null
for f in qkv proj fc2 attention nv12_batch; do python3 -c "
import json,sys; d=json.load(open('$O/r06_${f}_pmc.json')); print('$f', round(d['traffic_bytes_per_launch']/1e6,1), 'MB', round(d.get('traffic_over_algorithmic',0),3), 'x', 'L2 hit', round(d.get('l2_hit_rate',0),3), 'ns', round(d.get('kernel_ns_under_pmc',0)))"; done
