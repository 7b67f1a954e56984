# Round-end evidence run (GPU box): full -m gpu suite, default bench line, rocprofv3 kernel stats of the same command.
set -u
R=$GRAFT_REPO_ROOT
NAME=${1:-r01_i}
cd $R
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu > gpurun_out/${NAME}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${NAME}_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py --breakdown > gpurun_out/${NAME}_bench_default.json 2> gpurun_out/${NAME}_bench.err
rc=$?; cat gpurun_out/${NAME}_bench_default.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/${NAME}_bench.err; exit $rc; }
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${NAME}_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${NAME}_prof -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras > $R/gpurun_out/${NAME}_prof.log 2>&1
rc=$?; echo "rocprof rc=$rc"; [ $rc -ne 0 ] && exit $rc
cp $R/gpurun_out/${NAME}_prof/*/*kernel_stats.csv $R/gpurun_out/${NAME}_kernel_stats.csv
head -14 $R/gpurun_out/${NAME}_kernel_stats.csv | cut -c1-150
