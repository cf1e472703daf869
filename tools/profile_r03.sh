#!/bin/bash
# round-3 measurement set on one MI355X box (run from the repository root): rocprofv3 kernel-trace summaries of
# bench.py (one engine / two engines), the PMC passes of the encoder kernels, and the bench lines themselves.
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r03
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_30x1 -- python3 bench.py --steps 60 --warmup 10 --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_30x1.json 2> $O/trace_30x1.err
echo "trace 30x1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_60x2 -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_60x2.json 2> $O/trace_60x2.err
echo "trace 60x2 done"
bash tools/pmc_r03.sh > $O/pmc.log 2>&1
echo "pmc done"
python3 bench.py > $O/bench_cfg3_60x2.json 2> $O/bench_cfg3_60x2.err
echo "bench default done"
python3 bench.py --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/bench_cfg3_30x1.json 2>> $O/bench.err
python3 bench.py --workload cfg2 --no-cpu-baseline --no-host-leg --no-single-leg > $O/bench_cfg2.json 2>> $O/bench.err
python3 bench.py --workload cfg5 --steps 100 --no-cpu-baseline --no-host-leg --no-single-leg > $O/bench_cfg5.json 2>> $O/bench.err
echo "benches done"
find $O -name "*kernel_stats.csv" | head
