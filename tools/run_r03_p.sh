# GPU box, experiment P: (1) non-temporal policy on the reduce pass's streams (variant builds), per-kernel times;
# (2) reduce(group g) on a side stream beside bin(group g+1): whole-step rate per group count
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
bash tools/ab_kernels.sh $L/liblnerf_hip.so $L/liblnerf_hip_nt1.so $L/liblnerf_hip_nt3.so $L/liblnerf_hip_nt4.so $L/liblnerf_hip_nt7.so || exit 1
B="python3 bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for t in "" "--tune scatter_level_groups=2" "--tune scatter_level_groups=2,scatter_overlap=1" "--tune scatter_level_groups=4,scatter_overlap=1" "--tune scatter_level_groups=8,scatter_overlap=1" "--tune scatter_level_groups=16,scatter_overlap=1" ""; do
  echo "== $t"
  timeout -k 10 200 $B $t 2> gpurun_out/p_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/p_err.log; exit 1; }
done
