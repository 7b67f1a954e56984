# GPU box: same-box A/B of the round-2 tree (tools/r02_tree, exported from 0d6c8fe, its own library) against this tree,
# refresh-free step on both sides (round 2's bench had no refresh in its timed region), 4 interleaved rounds
set -u
R=$GRAFT_REPO_ROOT
cd $R
NAME=${1:-r03_ab}
for i in 1 2 3 4; do
  timeout -k 10 200 python3 tools/r02_tree/bench.py --no-cpu-baseline --no-extras > gpurun_out/${NAME}_old_$i.json 2> gpurun_out/${NAME}_old_$i.err || { tail -5 gpurun_out/${NAME}_old_$i.err; exit 1; }
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --refresh 0 > gpurun_out/${NAME}_new_$i.json 2> gpurun_out/${NAME}_new_$i.err || { tail -5 gpurun_out/${NAME}_new_$i.err; exit 1; }
done
python3 - $NAME <<'PY'
import json, glob, statistics, sys
for t in ("old", "new"):
    rows = [json.load(open(f)) for f in sorted(glob.glob("gpurun_out/%s_%s_*.json" % (sys.argv[1], t)))]
    print(json.dumps({"tree": "round 2 (0d6c8fe)" if t == "old" else "round 3 (this build, --refresh 0)", "build": rows[0]["build"], "runs": len(rows),
                      "fps_median": round(statistics.median(r["value"] for r in rows), 1), "fps_all": [round(r["value"], 1) for r in rows],
                      "scatter_ms_median": round(statistics.median(r["scatter"]["kernel_ms"] for r in rows), 4),
                      "gather_ms_median": round(statistics.median(r["roofline"]["kernel_ms"] for r in rows), 4)}))
PY
