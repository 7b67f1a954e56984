# GPU box, experiment Q: non-temporal policy matrix (reduce: params 3 / +records 7; bin: dfeat 1 / +record stores 3):
# whole-step rate (two interleaved rounds) and per-kernel times; then the overlapped group form with 2 bin WGs per CU
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
LIBS="liblnerf_hip.so liblnerf_hip_nt3.so liblnerf_hip_nt7.so liblnerf_hip_nt3b1.so liblnerf_hip_nt7b1.so liblnerf_hip_nt7b3.so"
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2; do
for lib in $LIBS; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/q_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'], d['roofline']['kernel_ms'])" || { tail -5 gpurun_out/q_err.log; exit 1; }
done
done
for t in "--tune scatter_bin_per_cu=2" "--tune scatter_bin_per_cu=2,scatter_level_groups=2,scatter_overlap=1" "--tune scatter_bin_per_cu=1,scatter_level_groups=2,scatter_overlap=1" "--tune scatter_bin_per_cu=2,scatter_level_groups=4,scatter_overlap=1"; do
  echo "== $t"
  timeout -k 10 200 $B $t 2> gpurun_out/q_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/q_err.log; exit 1; }
done
bash tools/ab_kernels.sh $L/liblnerf_hip.so $L/liblnerf_hip_nt7b1.so $L/liblnerf_hip_nt7b3.so $L/liblnerf_hip_nt3b1.so
