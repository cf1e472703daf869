set -e
cd /root/repo
python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_hardening.py tests/test_gpu_host.py -x -q 2>&1 | tail -3
for c in cfg3 cfg2; do echo "== tiers chosen per pass (round 5) $c"; python tools/preproc_bench.py $c; done 2>&1 | grep -v amdgpu
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg --no-single-leg --no-profile | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), round(d['whole_frame_mfma_frac'],4))"
