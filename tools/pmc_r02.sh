# PMC passes (one counter group per rocprofv3 run) of the round-2 kernels the verdict names beside fc1:
# proj and fc2 (f32 residual epilogue, 256x256 schedule v2), QKV (persistent) and attention mode 3.
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/pmc_r02b && mkdir -p $O && cd $R
bash tools/run_pmc.sh $O/proj python3 tools/one_gemm.py 21600 768 768 1 19 20 &&
bash tools/run_pmc.sh $O/fc2 python3 tools/one_gemm.py 21600 768 3072 1 19 20 &&
bash tools/run_pmc.sh $O/qkv python3 tools/one_gemm.py 21600 2304 768 4 19 20 &&
bash tools/run_pmc.sh $O/attn python3 tools/one_attn.py 30 3 20 &&
python tools/pmc_summary.py $O/proj "gemm256_kernel<1" $O/r02_proj_pmc.json --family gemm_bf16_resid_256x256pp_n768k768 --streams 30 --algorithmic-bytes $((21600*768*2 + 768*768*2 + 21600*768*8)) --command "python3 tools/one_gemm.py 21600 768 768 1 19 20" &&
python tools/pmc_summary.py $O/fc2 "gemm256_kernel<1" $O/r02_fc2_pmc.json --family gemm_bf16_resid_256x256pp_n768k3072 --streams 30 --algorithmic-bytes $((21600*3072*2 + 768*3072*2 + 21600*768*8)) --command "python3 tools/one_gemm.py 21600 768 3072 1 19 20" &&
python tools/pmc_summary.py $O/qkv "gemm256p_kernel<4" $O/r02_qkv_pmc.json --family gemm_bf16_qkv_256x256pp_n2304k768 --streams 30 --algorithmic-bytes $((21600*768*2 + 2304*768*2 + 21600*2304*2)) --command "python3 tools/one_gemm.py 21600 2304 768 4 19 20" &&
python tools/pmc_summary.py $O/attn "attention_dma_kernel" $O/r02_attention_pmc.json --family attention --streams 30 --algorithmic-bytes $((21600*768*2*4)) --command "python3 tools/one_attn.py 30 3 20"
ls $O/*.json
