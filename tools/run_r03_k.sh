# GPU box, round 3 step K: trainer (whole-step graph) + distributed tests, same-box A/B against the round-2 tree,
# PMC of the gather (hash vs blocked), default bench
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 900 python3 -m pytest tests/test_gpu_trainer.py tests/test_gpu_distributed.py -x -q -m gpu > gpurun_out/r03k_tests.log 2>&1
rc=$?; tail -5 gpurun_out/r03k_tests.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3 4; do
  timeout -k 10 200 python3 tools/r02_tree/bench.py --no-cpu-baseline --no-extras > gpurun_out/r03k_ab_old_$i.json 2> gpurun_out/r03k_ab_old_$i.err || { tail -5 gpurun_out/r03k_ab_old_$i.err; exit 1; }
  timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --refresh 0 > gpurun_out/r03k_ab_new_$i.json 2> gpurun_out/r03k_ab_new_$i.err || { tail -5 gpurun_out/r03k_ab_new_$i.err; exit 1; }
done
python3 - <<'PY'
import json, glob, statistics
for t in ("old", "new"):
    rows = [json.load(open(f)) for f in sorted(glob.glob("gpurun_out/r03k_ab_%s_*.json" % t))]
    print(json.dumps({"tree": "round 2 (0d6c8fe)" if t == "old" else "round 3 (this build, --refresh 0)", "build": rows[0]["build"], "runs": len(rows),
                      "fps_median": round(statistics.median(r["value"] for r in rows), 1), "fps_all": [round(r["value"], 1) for r in rows],
                      "scatter_ms_median": round(statistics.median(r["scatter"]["kernel_ms"] for r in rows), 4),
                      "gather_ms_median": round(statistics.median(r["roofline"]["kernel_ms"] for r in rows), 4)}))
PY
timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r03k_bench.json 2> gpurun_out/r03k_bench.err; python3 -c "
import json; d=json.load(open('gpurun_out/r03k_bench.json')); print('value', d['value'], 'no-refresh', d['refresh']['value_without_refresh']); print('trainer', d['trainer'])"
bash tools/run_pmc_all.sh r03k_pmc_hash > gpurun_out/r03k_pmc_hash.log 2>&1; tail -2 gpurun_out/r03k_pmc_hash.log | cut -c1-300
bash tools/run_pmc_all.sh r03k_pmc_blocked "--gridtype blocked" > gpurun_out/r03k_pmc_blocked.log 2>&1; tail -2 gpurun_out/r03k_pmc_blocked.log | cut -c1-300
python3 - <<'PY'
import json
for n in ("hash", "blocked"):
    d = json.load(open("gpurun_out/r03k_pmc_%s.json" % n))
    print(n, json.dumps(d.get("gather", {})))
PY
exit 0
