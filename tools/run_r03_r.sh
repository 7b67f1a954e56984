# GPU box, experiment R: non-temporal policy matrix, second round (record loads default; bin record stores nt)
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
LIBS="liblnerf_hip.so liblnerf_hip_nt3b1.so liblnerf_hip_nt7b3.so liblnerf_hip_nt3b3.so liblnerf_hip_nt3b2.so liblnerf_hip_nt1b3.so"
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
for lib in $LIBS; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/q_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'], d['roofline']['kernel_ms'])" || { tail -5 gpurun_out/q_err.log; exit 1; }
done
done
bash tools/ab_kernels.sh $L/liblnerf_hip.so $L/liblnerf_hip_nt3b3.so $L/liblnerf_hip_nt3b2.so $L/liblnerf_hip_nt1b3.so
