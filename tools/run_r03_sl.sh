# GPU box: records per slice workgroup of pass 2 (16384 / 32768 / 65536): fused step and the N > 1 step on one rank
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2; do
for lib in liblnerf_hip.so liblnerf_hip_sl32.so liblnerf_hip_sl64.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/sl_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/sl_err.log; exit 1; }
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B --force-dist 2> gpurun_out/sl_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('   force-dist $lib', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/sl_err.log; exit 1; }
done
done
