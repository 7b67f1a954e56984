# GPU box: stage 2 (pass 2 closes the step: slab sums + MLP Adam + tick inside the reduce launch): tests, timeline
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_render.py tests/test_gpu_trainer.py -x -q > gpurun_out/s2_tests.log 2>&1; rc=$?; tail -3 gpurun_out/s2_tests.log; [ $rc -ne 0 ] && exit $rc
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
  timeout -k 10 200 $B 2> gpurun_out/s2_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/s2_err.log; exit 1; }
  timeout -k 10 200 python3 tools/r02_tree/bench.py --no-cpu-baseline --no-extras 2> gpurun_out/s2_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('r02 tree', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/s2_err.log; exit 1; }
done
bash tools/run_trace.sh s2 > gpurun_out/s2_trace.log 2>&1; python3 tools/trace_timeline.py gpurun_out/s2_kernel_trace.csv
