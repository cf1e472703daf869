# rocprofv3 trace of the pipelined host ingest: tools/trace_host.sh <streams> <engines> <steps> <outdir>
cd /tmp && export TMPDIR=/tmp && R=$GRAFT_REPO_ROOT && O=$R/gpurun_out/$4 && mkdir -p $O &&
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O -- python3 $R/tools/host_pipelined.py $1 $2 $3 > $O.log 2>&1 &&
cd $R && python tools/trace_overlap.py $O 1048576 0 > $O.overlap.txt; cat $O.overlap.txt; tail -2 $O.log
