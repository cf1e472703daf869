set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "head_band or head_conv3x3" 2>&1 | tail -15
python -m pytest tests/test_gpu_pipeline.py -x -q -k "head_band_kernel_equals" 2>&1 | tail -15
python tools/headconv_bench.py 1,2,8,30 > gpurun_out/r5_headconv_sweep.txt 2>&1
cat gpurun_out/r5_headconv_sweep.txt
