set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "head_band" 2>&1 | tail -2
python -m pytest tests/test_gpu_pipeline.py -x -q -k "head_band" 2>&1 | tail -2
python tools/headconv_bench.py 1,30 2>&1 | grep -v amdgpu | cut -c1-200
