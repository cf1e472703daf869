cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out && mkdir -p $O && cd $R &&
python -m pytest tests -m gpu -x -q > $O/final_gpu_tests.log 2>&1 &&
python bench.py > $O/final_bench.json 2> $O/final_bench.err &&
cd /tmp &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof60 -o b60 -- python3 $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-host-leg > $O/prof60.log 2>&1 &&
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof30 -o b30 -- python3 $R/bench.py --steps 60 --warmup 10 --streams 30 --groups 1 --no-cpu-baseline --no-host-leg > $O/prof30.log 2>&1 &&
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/trace_host -- python3 $R/tools/host_pipelined.py 30 1 12 > $O/trace_host.log 2>&1 &&
cd $R && python tools/trace_overlap.py $O/trace_host 1048576 30000 > $O/overlap_gap30us.txt && python tools/trace_overlap.py $O/trace_host 1048576 0 > $O/overlap_gap0.txt;
tail -3 $O/final_gpu_tests.log; head -c 600 $O/final_bench.json; echo; cat $O/overlap_gap30us.txt $O/overlap_gap0.txt; ls $O/prof60 $O/prof30 | head; du -sh $O
