#!/bin/bash
# round-4 measurement set on one MI355X box (run from the repository root): rocprofv3 kernel-trace summaries of
# bench.py (one engine / two engines), the PMC passes of the encoder kernels, the bench lines, the full GPU test
# log with the parity prints, and the in-kernel phase stamps. Results under gpurun_out/r04/; the summaries that are
# judged are copied into profiles/ afterwards.
set -u
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
O=gpurun_out/r04
mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_30x1 -- python3 bench.py --steps 60 --warmup 10 --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_30x1.json 2> $O/trace_30x1.err
echo "trace 30x1 done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_60x2 -- python3 bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg --no-single-leg > $O/trace_60x2.json 2> $O/trace_60x2.err
echo "trace 60x2 done"
bash tools/pmc_r04.sh > $O/pmc.log 2>&1
echo "pmc done"
python3 bench.py > $O/bench_cfg3_60x2.json 2> $O/bench_cfg3_60x2.err
echo "bench default done"
python3 bench.py --streams 30 --groups 1 --no-cpu-baseline --no-host-leg --no-single-leg > $O/bench_cfg3_30x1.json 2>> $O/bench.err
python3 bench.py --workload cfg2 --no-cpu-baseline --no-single-leg > $O/bench_cfg2.json 2>> $O/bench.err
python3 bench.py --workload cfg5 --steps 100 --no-cpu-baseline --no-single-leg > $O/bench_cfg5.json 2>> $O/bench.err
echo "benches done"
find $O -name "*kernel_stats.csv" | head
if [ "${SKIP_TESTS:-0}" != "1" ]; then python3 -m pytest tests -m gpu -q -s > $O/r04_gpu_tests.log 2>&1; fi
echo "gpu tests rc $?"
export VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_stamps.so
if [ -f $VITTRACK_HIP_LIB ]; then
{ echo "# build.py --stamps; persistent kernel: [top-of-tile wait, epilogue, main loop, total]; X-epilogue kernel: [main loop, epilogue, statistics hand-off, total]; cycles per wave";
  for a in "21600 3072 768 2 19" "21600 2304 768 4 19" "21600 768 3072 1 18" "21600 768 768 1 18"; do echo "one_gemm $a 20:"; python3 tools/one_gemm.py $a 20 2>&1 | grep -v amdgpu; done; } > $O/r04_gemm_phase_stamps.txt
fi
unset VITTRACK_HIP_LIB
echo "stamps done"
