# GPU box, round 3 step A: new host-side paths (graphed trainer, forced RCCL exchange, bench with refresh + trainer)
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_trainer.py tests/test_gpu_distributed.py tests/test_gpu_render.py -x -q -m gpu > gpurun_out/r03a_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r03a_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py > gpurun_out/r03a_bench.json 2> gpurun_out/r03a_bench.err; rc=$?; tail -c 2500 gpurun_out/r03a_bench.json; [ $rc -ne 0 ] && { tail -20 gpurun_out/r03a_bench.err; exit $rc; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03a_bench20.json 2> gpurun_out/r03a_bench20.err; tail -c 1200 gpurun_out/r03a_bench20.json
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/r03a_bench_forcedist.json 2> gpurun_out/r03a_bench_forcedist.err; rc=$?; tail -c 1500 gpurun_out/r03a_bench_forcedist.json; [ $rc -ne 0 ] && tail -20 gpurun_out/r03a_bench_forcedist.err
exit 0
