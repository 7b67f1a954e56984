# GPU box: bench.py under several lnerf_set_tuning settings, alternated (3 rounds).  usage: tools/tune_sweep.sh "k=v,k=v" "k=v" ...
# ("none" = no override)
set -u
i=0
for r in 1 2 3; do i=0; for t in "$@"; do i=$((i+1))
  if [ "$t" = "none" ]; then T=""; else T="--tune $t"; fi
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --refresh 0 $T > gpurun_out/sw_${i}_$r.json 2>/dev/null || echo "fail $t"
done; done
python3 - "$@" <<'PY'
import json, sys, statistics, glob
for i, t in enumerate(sys.argv[1:], 1):
    rows = [json.load(open(f)) for f in glob.glob("gpurun_out/sw_%d_*.json" % i)]
    if rows:
        print(json.dumps({"tune": t, "fps_median": round(statistics.median(r["value"] for r in rows), 1),
                          "gather_ms": round(statistics.median(r["roofline"]["kernel_ms"] for r in rows), 4),
                          "scatter_ms": round(statistics.median(r["scatter"]["kernel_ms"] for r in rows), 4),
                          "mlp_bwd_ms": round(statistics.median(r["mfma"]["bwd_ms"] for r in rows), 4),
                          "mlp_fwd_ms": round(statistics.median(r["mfma"]["fwd_ms"] for r in rows), 4)}))
PY
