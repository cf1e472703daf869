#!/bin/bash
# shader clock and socket power while the hot path runs: rocm-smi samples beside a long bench.py run
# (bash tools/clock_under_load.sh > gpurun_out/r03/clock_under_load.txt)
set -u
rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|power" | head -4
echo "--- idle above, under load below (bench.py --steps 600, 60 streams x 2 engines) ---"
python3 bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-host-leg --no-single-leg > /tmp/clk_bench.json 2>/dev/null &
BP=$!
sleep 25      # import, weights, warm-up
for i in 1 2 3 4 5 6; do
    rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Power\|Average Graphics" | tr '\n' ' '; echo
    sleep 0.7
done
wait $BP
python3 tools/show_bench.py /tmp/clk_bench.json | head -2
