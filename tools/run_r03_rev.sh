# GPU box: work-unit order of pass 2 (level order / fine levels first): step rate, kernel time, a parity subset
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
for lib in liblnerf_hip_rev0.so liblnerf_hip_rev1.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/rev_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/rev_err.log; exit 1; }
done
done
bash tools/ab_kernels.sh $L/liblnerf_hip_rev0.so $L/liblnerf_hip_rev1.so | grep -v "k_mlp\|k_grid_forward"
LNERF_HIP_LIB=$R/$L/liblnerf_hip_rev1.so timeout -k 10 600 python3 -m pytest tests/test_gpu_render.py -x -q -k "fused or tail or full_size" > gpurun_out/rev_tests.log 2>&1; tail -2 gpurun_out/rev_tests.log
