# GPU box, run Z: fused reduce pass with 512-thread workgroups (U = 8 / 16) against 1024
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2; do
for lib in liblnerf_hip.so liblnerf_hip_u6.so liblnerf_hip_u4.so liblnerf_hip_u2.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/z_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/z_err.log; exit 1; }
done
done
bash tools/ab_kernels.sh $L/liblnerf_hip.so $L/liblnerf_hip_u6.so $L/liblnerf_hip_u4.so $L/liblnerf_hip_u2.so
