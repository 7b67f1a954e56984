# GPU box, round 3 step D: reduce pass with one segment per round (variants U = 4 / 8 / 16), parity, PMC of the scatter
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or scatter" > gpurun_out/r03d_parity.log 2>&1
rc=$?; tail -5 gpurun_out/r03d_parity.log; [ $rc -ne 0 ] && exit $rc
bash tools/ab_kernels.sh $L/liblnerf_hip_r02scatter.so $L/liblnerf_hip.so $L/liblnerf_hip_u4.so $L/liblnerf_hip_u16.so > gpurun_out/r03d_abk.txt 2>&1; cat gpurun_out/r03d_abk.txt | grep -v "k_mlp\|k_grid_forward"
bash tools/ab_bench.sh 3 $L/liblnerf_hip_r02scatter.so $L/liblnerf_hip.so > gpurun_out/r03d_ab.jsonl 2>&1; cat gpurun_out/r03d_ab.jsonl
bash tools/run_pmc_scatter.sh r03d_pmc_scatter > gpurun_out/r03d_pmc.log 2>&1; tail -5 gpurun_out/r03d_pmc.log
timeout -k 10 900 python3 -m pytest tests/test_gpu_distributed.py tests/test_gpu_render.py tests/test_gpu_golden.py -x -q -m gpu -s > gpurun_out/r03d_tests.log 2>&1
rc=$?; tail -8 gpurun_out/r03d_tests.log; grep "full-size bf16" gpurun_out/r03d_tests.log
exit 0
