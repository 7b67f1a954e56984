# GPU box: last-arriver tile sums batched (all of a lane's elements of a tile in flight) + fine levels first:
# parity subset, fused step and N > 1 step against the committed build's library (rev0 = same source, plain order)
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py tests/test_gpu_distributed.py -x -q > gpurun_out/la_tests.log 2>&1; rc=$?; tail -2 gpurun_out/la_tests.log; [ $rc -ne 0 ] && exit $rc
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
for lib in liblnerf_hip_rev0.so liblnerf_hip.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/la_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/la_err.log; exit 1; }
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B --force-dist 2> gpurun_out/la_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   force-dist $lib', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/la_err.log; exit 1; }
done
done
