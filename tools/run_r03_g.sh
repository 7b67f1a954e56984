# GPU box, round 3 step G: reduce pass, one segment per round, branch-free loads (U = 8 default, U = 4)
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid or scatter" > gpurun_out/r03g_parity.log 2>&1
rc=$?; tail -3 gpurun_out/r03g_parity.log; [ $rc -ne 0 ] && exit $rc
bash tools/ab_kernels.sh $L/liblnerf_hip_r02scatter.so $L/liblnerf_hip.so $L/liblnerf_hip_u8.so > gpurun_out/r03g_abk.txt 2>&1; cat gpurun_out/r03g_abk.txt | grep -v "k_mlp\|k_grid_forward"
bash tools/ab_bench.sh 3 $L/liblnerf_hip_r02scatter.so $L/liblnerf_hip.so $L/liblnerf_hip_u8.so > gpurun_out/r03g_ab.jsonl 2>&1; cat gpurun_out/r03g_ab.jsonl
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r03g_bench.json 2> gpurun_out/r03g_bench.err; python3 -c "
import json; d=json.load(open('gpurun_out/r03g_bench.json')); print(d['value'], d['refresh']['value_without_refresh'], d['trainer'])"
exit 0
