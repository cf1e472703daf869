set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "head_band" 2>&1 | tail -2
python -m pytest tests/test_gpu_pipeline.py -x -q -k "head_band" 2>&1 | tail -2
python tools/headconv_bench.py 1,30 2>&1 | grep -v amdgpu | cut -c1-200
echo "== HSA_ENABLE_SDMA=0 (copies by blit kernels instead of the SDMA engines)"
HSA_ENABLE_SDMA=0 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-single-leg --no-profile > gpurun_out/r5_h2d_d.json 2> gpurun_out/r5_h2d_d.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_h2d_d.json'))
f=d['full_frame']
print('d value', round(d['value']), 'full_frame', round(f['value']), 'h2d GB/s', round(f['h2d_GBps'],1), 'alone', round(f['h2d_alone_GBps'],1), 'zero_copy', round(d['zero_copy']['value']), 'host_sync', round(d['host_synchronous']['value']))
PY
