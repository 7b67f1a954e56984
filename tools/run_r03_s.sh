# GPU box, experiment S: non-temporal feature loads in the MLP backward (last use), against the new default policy
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
LIBS="liblnerf_hip_nt0.so liblnerf_hip.so liblnerf_hip_mlpnt1.so"
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
for lib in $LIBS; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/q_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'], d['roofline']['kernel_ms'], d['mfma']['bwd_ms'])" || { tail -5 gpurun_out/q_err.log; exit 1; }
done
done
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py tests/test_gpu_golden.py -x -q > gpurun_out/s_tests.log 2>&1; tail -3 gpurun_out/s_tests.log
