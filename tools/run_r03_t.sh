# GPU box, run T: the exchange captured into the step graph (RCCL, one rank): tests + bench --force-dist with and without
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 800 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_trainer.py -x -q > gpurun_out/t_tests.log 2>&1; rc=$?; tail -3 gpurun_out/t_tests.log; [ $rc -ne 0 ] && exit $rc
for gc in 1 0 1 0; do
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline --no-extras --graph-collectives $gc 2> gpurun_out/t_fd$gc.err | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('graph-collectives $gc', d['value'], d['ms_per_step'], d['host_enqueue_ms_per_step'], d['launch'])" || { tail -5 gpurun_out/t_fd$gc.err; exit 1; }
done
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras 2> /dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('fused', d['value'], d['ms_per_step'])"
