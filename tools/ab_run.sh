python -m pytest tests -m gpu -x -q 2>&1 | tail -2
for s in 1 2 3; do python bench.py --no-cpu-baseline --no-host-leg --no-profile --streams $s --groups 1 --steps 200 --warmup 30 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('streams', d['config']['streams_per_gpu'], round(d['value'],1), round(d['ms_per_step'],4), d['tracked_ok'])"; done
python tools/gemm_ab.py 0,2,4,5 1,2 5
