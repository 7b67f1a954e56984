# GPU box, run V: bitmap walk in the reduce pass against the scalar walk (variant build walk0): parity tests, kernel times,
# whole-step rate (fused and --force-dist)
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py tests/test_gpu_golden.py -x -q > gpurun_out/v_tests.log 2>&1; rc=$?; tail -3 gpurun_out/v_tests.log; [ $rc -ne 0 ] && exit $rc
L=latent-nerf-test_amd/lib
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
for lib in liblnerf_hip_walk0.so liblnerf_hip.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/v_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'], d['roofline']['kernel_ms'])" || { tail -5 gpurun_out/v_err.log; exit 1; }
done
done
for lib in liblnerf_hip_walk0.so liblnerf_hip.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B --force-dist 2> gpurun_out/v_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('force-dist $lib', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/v_err.log; exit 1; }
done
bash tools/ab_kernels.sh $L/liblnerf_hip_walk0.so $L/liblnerf_hip.so
