# GPU box, round 3 step I: full -m gpu suite on the item-chunk scatter + step tail, A/B of the tail, kernel timeline
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/r03i_tests.log 2>&1
rc=$?; tail -12 gpurun_out/r03i_tests.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  for t in 0 1; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --tail $t > gpurun_out/r03i_tail${t}_$i.json 2> gpurun_out/r03i_tail${t}_$i.err || exit 1
  done
done
python3 - <<'PY'
import json, glob, statistics
for t in (0, 1):
    rows = [json.load(open(f)) for f in sorted(glob.glob("gpurun_out/r03i_tail%d_*.json" % t))]
    print(json.dumps({"tail": t, "fps_median": round(statistics.median(r["value"] for r in rows), 1), "ms_per_step": round(statistics.median(r["ms_per_step"] for r in rows), 4)}))
PY
bash tools/run_trace.sh r03i > gpurun_out/r03i_trace.log 2>&1; tail -3 gpurun_out/r03i_trace.log
python3 tools/trace_timeline.py gpurun_out/r03i_kernel_trace.csv > gpurun_out/r03i_step_timeline.json 2>gpurun_out/r03i_timeline.err; head -c 2500 gpurun_out/r03i_step_timeline.json

timeout -k 10 300 python3 bench.py --no-cpu-baseline > gpurun_out/r03i_bench.json 2> gpurun_out/r03i_bench.err; python3 -c "
import json; d=json.load(open('gpurun_out/r03i_bench.json')); print('value', d['value'], 'no-refresh', d['refresh']['value_without_refresh']); print('trainer', d['trainer']); print('blocked', d['blocked']); print('gather', d['roofline']['kernel_ms'], 'scatter', d['scatter']['kernel_ms'])"
exit 0
