# GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of a short bench run for several library builds.
# usage: tools/ab_kernels.sh lib1.so lib2.so ...   -> gpurun_out/abk_<n>_kernel_stats.csv + a table on stdout
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
k=0
for lib in "$@"; do
  k=$((k+1))
  rm -rf $R/gpurun_out/abk_$k
  LNERF_HIP_LIB=$R/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/abk_$k -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --refresh 0 > $R/gpurun_out/abk_$k.log 2>&1 || { tail -5 $R/gpurun_out/abk_$k.log; exit 1; }
  cp $R/gpurun_out/abk_$k/*/*kernel_stats.csv $R/gpurun_out/abk_${k}_kernel_stats.csv
  echo "== $lib"
  python3 - $R/gpurun_out/abk_${k}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(t in n for t in ("k_scatter", "k_grid_forward", "k_mlp_backward_bf16<", "k_mlp_forward_bf16")):
        print("  %-60s calls %5s avg_us %8.2f" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
