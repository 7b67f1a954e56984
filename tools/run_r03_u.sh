# GPU box, run U: the un-fused reduce pass (bf16 gradient output, no Adam) as ONE launch: is the record phase bound by HBM?
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in 1 4; do
rm -rf $R/gpurun_out/u_$g
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/u_$g -- python3 $R/bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extras --refresh 0 --force-dist --exchange-groups $g > $R/gpurun_out/u_$g.log 2>&1 || { tail -5 $R/gpurun_out/u_$g.log; exit 1; }
echo "== exchange-groups $g"
python3 - $R/gpurun_out/u_$g/*/*kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(t in n for t in ("k_scatter", "k_adam", "k_grid_forward")):
        print("  %-60s calls %5s avg_us %8.2f" % (n.split("(")[0][-60:], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
