set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "attention" 2>&1 | tail -3
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_host.py tests/test_gpu_threading.py -x -q 2>&1 | tail -3
python tools/attn_ab.py 3,7,8,9,10,5 30 7 > gpurun_out/r5_attn_ab.txt 2>&1
cat gpurun_out/r5_attn_ab.txt
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg > gpurun_out/r5_bench_b.json 2> gpurun_out/r5_bench_b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_b.json'))
print(d['value'], d['whole_frame_mfma_frac'], d['device_only']['value'])
for k in d['kernels']: print(k)
s=d['single_stream']; print({k:v for k,v in s.items() if k!='kernels'})
for k in s['kernels']: print(k)
PY
