# GPU box, round 3 step B: item-chunk scatter -- parity, trainer / distributed tests, same-box A/B against round 2's scatter
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q -m gpu > gpurun_out/r03b_parity.log 2>&1
rc=$?; tail -15 gpurun_out/r03b_parity.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 tools/bench_scatter.py --check --configs 3:0 > gpurun_out/r03b_scatter.json 2> gpurun_out/r03b_scatter.err; rc=$?; cat gpurun_out/r03b_scatter.json; [ $rc -ne 0 ] && { tail -5 gpurun_out/r03b_scatter.err; exit $rc; }
bash tools/ab_bench.sh 3 latent-nerf-test_amd/lib/liblnerf_hip_r02scatter.so latent-nerf-test_amd/lib/liblnerf_hip.so > gpurun_out/r03b_ab.jsonl 2>&1; cat gpurun_out/r03b_ab.jsonl
bash tools/ab_kernels.sh latent-nerf-test_amd/lib/liblnerf_hip_r02scatter.so latent-nerf-test_amd/lib/liblnerf_hip.so > gpurun_out/r03b_abk.txt 2>&1; tail -30 gpurun_out/r03b_abk.txt
timeout -k 10 900 python3 -m pytest tests/test_gpu_trainer.py tests/test_gpu_distributed.py tests/test_gpu_render.py -x -q -m gpu > gpurun_out/r03b_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r03b_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python3 bench.py > gpurun_out/r03b_bench.json 2> gpurun_out/r03b_bench.err; rc=$?; tail -c 3000 gpurun_out/r03b_bench.json; [ $rc -ne 0 ] && { tail -20 gpurun_out/r03b_bench.err; exit $rc; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03b_bench20.json 2> gpurun_out/r03b_bench20.err; tail -c 1500 gpurun_out/r03b_bench20.json
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/r03b_bench_forcedist.json 2> gpurun_out/r03b_bench_forcedist.err; rc=$?; tail -c 1500 gpurun_out/r03b_bench_forcedist.json; [ $rc -ne 0 ] && tail -20 gpurun_out/r03b_bench_forcedist.err
exit 0
