cd /root/repo
for cfg in "60 2" "120 2" "90 2" "60 1" "118 2"; do set -- $cfg
python bench.py --streams $1 --groups $2 --steps 40 --warmup 8 --no-cpu-baseline --no-host-leg --no-single-leg --no-profile > gpurun_out/r5_sweep_$1x$2.json 2> gpurun_out/r5_sweep_$1x$2.err
python - "$1" "$2" <<'PY'
import json,sys
d=json.load(open(f'gpurun_out/r5_sweep_{sys.argv[1]}x{sys.argv[2]}.json'))
print(sys.argv[1], sys.argv[2], 'value', round(d['value']), 'frac', round(d['whole_frame_mfma_frac'],4), 'device_only', round(d['device_only']['value']), 'ms/step', round(d['ms_per_step'],2))
PY
done
