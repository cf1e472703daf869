#!/bin/bash
# Sample the GPU's power, clocks and temperature (rocm-smi) while a command runs: is the pass held by the power / clock management?
#   bash tools/power_probe.sh OUTFILE -- command...
out=$1; shift; shift
"$@" > $out.cmd.log 2>&1 &
pid=$!
sleep 12       # library load, weights, warm-up
for i in 1 2 3 4 5 6 7 8; do
    if ! kill -0 $pid 2>/dev/null; then break; fi
    rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)|GPU use" | tr -s ' ' | head -12 >> $out
    echo "--" >> $out
    sleep 1
done
wait $pid
