# GPU box: bench.py with several builds of the library, alternately, in ONE call (boxes of the pool differ by a few %
# and a run by +-2 %, so an A/B across calls says little).  usage: tools/ab_bench.sh ROUNDS lib1.so lib2.so ...
set -u
N=$1; shift
for i in $(seq $N); do
  k=0
  for lib in "$@"; do
    k=$((k+1))
    LNERF_HIP_LIB=$lib timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras > gpurun_out/ab_${k}_$i.json 2> gpurun_out/ab_${k}_$i.err || exit 1
  done
done
python3 - "$@" <<'PY'
import glob, json, statistics, sys
for k, lib in enumerate(sys.argv[1:], 1):
    rows = [json.load(open(f)) for f in sorted(glob.glob("gpurun_out/ab_%d_*.json" % k))]
    v = [r["value"] for r in rows]; s = [r["scatter"]["kernel_ms"] for r in rows]
    print(json.dumps({"lib": lib.split("/")[-1], "build": rows[0]["build"], "runs": len(v), "fps_median": round(statistics.median(v), 1),
                      "fps_max": round(max(v), 1), "scatter_ms_median": round(statistics.median(s), 4), "scatter_ms_min": round(min(s), 4)}))
PY
