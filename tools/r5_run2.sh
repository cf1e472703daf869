set -e
cd /root/repo
python -m pytest tests/test_gpu_ops.py -x -q -k "head_band or head_conv3x3" 2>&1 | tail -3
python tools/headconv_bench.py 1,30 > gpurun_out/r5_headconv_sweep2.txt 2>&1
cat gpurun_out/r5_headconv_sweep2.txt
python -m pytest tests -m gpu -x -q 2>&1 | tail -15
python bench.py --steps 100 --warmup 20 > gpurun_out/r5_bench_a.json 2> gpurun_out/r5_bench_a.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5_bench_a.json'))
print(d['value'], d['whole_frame_mfma_frac'], d['device_only']['value'])
for k in d['kernels']: print(k)
s=d['single_stream']; print({k:v for k,v in s.items() if k!='kernels'})
for k in s['kernels']: print(k)
PY
