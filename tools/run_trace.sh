# GPU box: kernel trace (start/end timestamps) of a short bench run, for per-step timelines (tools/trace_timeline.py)
set -u
R=$GRAFT_REPO_ROOT
NAME=${1:-trace}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${NAME}_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${NAME}_prof -- python3 $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-extras --refresh 0 ${2:-} > $R/gpurun_out/${NAME}_prof.log 2>&1
rc=$?; echo "rocprof rc=$rc"; [ $rc -ne 0 ] && exit $rc
cp $R/gpurun_out/${NAME}_prof/*/*kernel_stats.csv $R/gpurun_out/${NAME}_kernel_stats.csv
cp $R/gpurun_out/${NAME}_prof/*/*kernel_trace.csv $R/gpurun_out/${NAME}_kernel_trace.csv
ls -la $R/gpurun_out/${NAME}_kernel_trace.csv
rm -rf $R/gpurun_out/${NAME}_prof
