set -e
cd /root/repo
python -m pytest tests/test_gpu_pipeline.py -x -q 2>&1 | tail -3
python -m pytest tests/test_gpu_trajectories.py -x -q -k "bench_times" -s 2>&1 | tail -8
for c in cfg3 cfg2; do
  echo "== new (strips) $c"; python tools/preproc_bench.py $c
  echo "== old $c"; VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_oldpre.so python tools/preproc_bench.py $c
done > gpurun_out/r5_preproc_by_target.txt 2>&1
cat gpurun_out/r5_preproc_by_target.txt
python bench.py --streams 90 --groups 3 --steps 40 --warmup 8 --no-cpu-baseline --no-host-leg --no-single-leg --no-profile > gpurun_out/r5_bench_3x30.json 2> gpurun_out/r5_bench_3x30.err
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg --no-single-leg --no-profile > gpurun_out/r5_bench_2x30.json 2> gpurun_out/r5_bench_2x30.err
python - <<'PY'
import json
for n in ("3x30","2x30"):
    d=json.load(open(f'gpurun_out/r5_bench_{n}.json'))
    print(n, d['value'], d['whole_frame_mfma_frac'], d['device_only']['value'])
PY
