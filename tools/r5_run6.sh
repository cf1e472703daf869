set -e
cd /root/repo
python tools/attn_ab.py 0,1,2,3,6 1 7 > gpurun_out/r5_attn_ab_b1.txt 2>&1; cat gpurun_out/r5_attn_ab_b1.txt
python tools/attn_ab.py 0,2,3 2 5 2>&1 | grep mode
echo "== full_frame leg with the engines' own copy streams never created (--ingest device): 2 compute + 1 copy + default stream"
python bench.py --ingest device --steps 40 --warmup 8 --no-cpu-baseline --no-single-leg --no-profile > gpurun_out/r5_h2d_a.json 2> gpurun_out/r5_h2d_a.err
echo "== default order (headline host ingest first: 2 compute + 2 engine copy streams + bench copy stream + default)"
python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-single-leg --no-profile > gpurun_out/r5_h2d_b.json 2> gpurun_out/r5_h2d_b.err
echo "== default order with GPU_MAX_HW_QUEUES=8"
GPU_MAX_HW_QUEUES=8 python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-single-leg --no-profile > gpurun_out/r5_h2d_c.json 2> gpurun_out/r5_h2d_c.err
python - <<'PY'
import json
for n in "abc":
    d=json.load(open(f'gpurun_out/r5_h2d_{n}.json'))
    f=d['full_frame']
    print(n, 'value', round(d['value']), 'full_frame', round(f['value']), 'h2d GB/s', round(f['h2d_GBps'],1), 'alone', round(f['h2d_alone_GBps'],1), 'zero_copy', round(d['zero_copy']['value']), 'host_sync', round(d['host_synchronous']['value']), d.get('pcie_inclusive'))
PY
