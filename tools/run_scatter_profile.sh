# GPU box: parity tests of the hash-grid scatter, then the scatter of the bench step timed per configuration
# (tools/bench_scatter.py) and its kernels under rocprofv3 --kernel-trace --stats.
set -u
R=$GRAFT_REPO_ROOT
NAME=${1:-r02_scatter}
CONFIGS=${2:-3:0,2:0}
cd $R
timeout -k 10 120 python3 tools/dbg_scatter.py > gpurun_out/${NAME}_dbg.log 2>&1; grep -c "eq01 True eq02 True" gpurun_out/${NAME}_dbg.log; grep "False" gpurun_out/${NAME}_dbg.log
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu -k "grid or scatter or golden" > gpurun_out/${NAME}_tests.log 2>&1
rc=$?; tail -15 gpurun_out/${NAME}_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 tools/bench_scatter.py --check --configs $CONFIGS > gpurun_out/${NAME}_bench.json 2> gpurun_out/${NAME}_bench.err
rc=$?; cat gpurun_out/${NAME}_bench.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/${NAME}_bench.err; exit $rc; }
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/${NAME}_prof
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${NAME}_prof -- python3 $R/tools/bench_scatter.py --rounds 10 --configs $CONFIGS > $R/gpurun_out/${NAME}_prof.log 2>&1
rc=$?; echo "rocprof rc=$rc"; [ $rc -ne 0 ] && exit $rc
cp $R/gpurun_out/${NAME}_prof/*/*kernel_stats.csv $R/gpurun_out/${NAME}_kernel_stats.csv
grep "scatter" $R/gpurun_out/${NAME}_kernel_stats.csv | cut -c1-60,200-400
