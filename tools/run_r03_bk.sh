# GPU box: 2048-row buckets (32 KiB of accumulators; four 512-thread workgroups per CU in the fused pass) against 4096
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
LNERF_HIP_LIB=$R/$L/liblnerf_hip_bk11.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py -x -q -k "grid_encode or scatter or fused or tail or full_size" > gpurun_out/bk_tests.log 2>&1; tail -3 gpurun_out/bk_tests.log
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2; do
for lib in liblnerf_hip.so liblnerf_hip_bk11.so liblnerf_hip_bk11rt1024.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/bk_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/bk_err.log; }
done
done
bash tools/ab_kernels.sh $L/liblnerf_hip.so $L/liblnerf_hip_bk11.so $L/liblnerf_hip_bk11rt1024.so
