#!/bin/bash
# samples board power / clocks (rocm-smi, ordinary user) while bench.py's timed loop runs: is the pass power-limited?
OUT=gpurun_out/power_probe; mkdir -p $OUT
rocm-smi --showpower --showmaxpower --showclocks > $OUT/idle.txt 2>&1
python bench.py --steps 600 --warmup 50 --no-cpu-baseline --no-host-leg --no-single-leg --no-profile --ingest device > $OUT/bench.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 1 12); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use" ; sleep 0.4; done > $OUT/busy.txt
wait $BP
cat $OUT/idle.txt | grep -E "Power|sclk|Max" ; echo ----; cat $OUT/busy.txt | sort | uniq -c | sort -rn | head -30
python3 -c "
import json;d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]);print(d['value'],d['ms_per_step'])"
