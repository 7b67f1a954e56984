# GPU box, round 3 step C: reduce-pass variants of the item-chunk scatter (same-box A/B), remaining tests, bench lines
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
bash tools/ab_kernels.sh $L/liblnerf_hip_r02scatter.so $L/liblnerf_hip.so $L/liblnerf_hip_u8.so $L/liblnerf_hip_noxcd.so > gpurun_out/r03c_abk.txt 2>&1; cat gpurun_out/r03c_abk.txt | grep -v "k_mlp\|k_grid_forward"
bash tools/ab_bench.sh 3 $L/liblnerf_hip_r02scatter.so $L/liblnerf_hip.so $L/liblnerf_hip_u8.so > gpurun_out/r03c_ab.jsonl 2>&1; cat gpurun_out/r03c_ab.jsonl
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_render.py tests/test_gpu_golden.py -x -q -m gpu -s > gpurun_out/r03c_tests.log 2>&1
rc=$?; tail -25 gpurun_out/r03c_tests.log; grep "full-size bf16" gpurun_out/r03c_tests.log
timeout -k 10 300 python3 bench.py > gpurun_out/r03c_bench.json 2> gpurun_out/r03c_bench.err; rc=$?; tail -c 3000 gpurun_out/r03c_bench.json; [ $rc -ne 0 ] && { tail -20 gpurun_out/r03c_bench.err; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/r03c_bench20.json 2> gpurun_out/r03c_bench20.err; tail -c 1500 gpurun_out/r03c_bench20.json
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/r03c_bench_forcedist.json 2> gpurun_out/r03c_bench_forcedist.err; rc=$?; tail -c 1500 gpurun_out/r03c_bench_forcedist.json; [ $rc -ne 0 ] && tail -20 gpurun_out/r03c_bench_forcedist.err
exit 0
