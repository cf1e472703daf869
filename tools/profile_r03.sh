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
# parity report (prints of the closed-loop, teacher-forced and stage-tap tests), single stream, in-kernel stamps
python3 -m pytest tests/test_gpu_trajectories.py tests/test_gpu_pipeline.py -q -s -k "traject or closed_loop or teacher or taps or large_engine" > $O/r03_parity_report.txt 2>&1
echo "parity report done"
python3 bench.py --streams 1 --groups 1 --no-cpu-baseline --no-host-leg > $O/bench_cfg3_single_stream.json 2>> $O/bench.err
export VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_clk.so
{ echo "# build.py --variant clk -DVT_STAMPS -DVT_STAMPS_CLOCK; per wave: [s_memrealtime ticks (100 MHz), epilogue | hand-off, main loop | epilogue, total] s_memtime cycles";
  for a in "21600 3072 768 2 19" "21600 2304 768 4 19" "21600 768 3072 1 18" "21600 768 768 1 18"; do echo "one_gemm $a 20:"; python3 tools/one_gemm.py $a 20 2>&1 | grep -v amdgpu; done; } > $O/r03_gemm_clock_stamps.txt
export VITTRACK_HIP_LIB=$PWD/gstreamer-vit-tracker_amd/libvittrack_hip_stamps.so
{ echo "# build.py --stamps; persistent kernel: [top-of-tile wait, epilogue, main loop, total]; X-epilogue kernel: [main loop, epilogue, statistics hand-off, total]; cycles per wave";
  for a in "21600 3072 768 2 19" "21600 2304 768 4 19" "21600 768 3072 1 18" "21600 768 768 1 18" "21600 3072 768 2 16"; do echo "one_gemm $a 20:"; python3 tools/one_gemm.py $a 20 2>&1 | grep -v amdgpu; done; } > $O/r03_gemm_phase_stamps.txt
unset VITTRACK_HIP_LIB
hipcc --offload-arch=gfx950 -O3 -w tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak > $O/r03_mfma_peak.txt 2>&1
echo "stamps done"
