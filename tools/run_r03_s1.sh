# GPU box: stage 1 (sliced buckets finished by the last slice inside pass 2): tests, step rate, kernel times
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py tests/test_gpu_golden.py tests/test_gpu_trainer.py -x -q > gpurun_out/s1_tests.log 2>&1; rc=$?; tail -3 gpurun_out/s1_tests.log; [ $rc -ne 0 ] && exit $rc
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
  timeout -k 10 200 $B 2> gpurun_out/s1_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/s1_err.log; exit 1; }
done
bash tools/run_trace.sh s1 > gpurun_out/s1_trace.log 2>&1; python3 tools/trace_timeline.py gpurun_out/s1_kernel_trace.csv
