#!/bin/bash
# power and clock of the register-only MFMA loop (tools/mfma_power.hip), rocm-smi sampled beside it:  bash tools/mfma_power.sh
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/mfma_power.hip -o /tmp/mfma_power 2>/dev/null || exit 1
for cfg in "0 1 5 2" "1 1 5 2" "0 0 5 2" "1 0 5 2" "0 1 5 1" "1 1 5 1"; do
    /tmp/mfma_power $cfg > /tmp/mfma_power.out &
    pid=$!
    sleep 2
    pw=""; sc=""
    for i in 1 2 3 4; do
        s=$(rocm-smi --showpower --showclocks 2>/dev/null)
        pw="$pw $(echo "$s" | grep -E "Power" | grep -oE "[0-9]+\.[0-9]+" | head -1)"
        sc="$sc $(echo "$s" | grep -E "sclk" | grep -oE "\([0-9]+Mhz\)" | head -1)"
        sleep 0.4
    done
    wait $pid
    echo "$(cat /tmp/mfma_power.out)   power [W]:$pw   sclk:$sc"
done
