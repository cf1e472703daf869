set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "head_band or head_conv3x3" 2>&1 | tail -3
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_threading.py tests/test_gpu_hardening.py -x -q 2>&1 | tail -3
python tools/headconv_bench.py 1,30 > gpurun_out/r5_headconv_sweep3.txt 2>&1
cat gpurun_out/r5_headconv_sweep3.txt
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg > gpurun_out/r5_bench_c.json 2> gpurun_out/r5_bench_c.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_c.json'))
print(d['value'], d['whole_frame_mfma_frac'], d['device_only']['value'])
for k in d['kernels'][5:]: print(k)
s=d['single_stream']; print({k:v for k,v in s.items() if k!='kernels'})
for k in s['kernels'][5:]: print(k)
PY
