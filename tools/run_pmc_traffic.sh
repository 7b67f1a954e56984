# HBM-side traffic of every kernel of one eager bench step (FETCH_SIZE / WRITE_SIZE in KiB, one counter per run).
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_traffic_$c
  timeout -k 10 120 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_traffic_$c -- python3 $R/bench.py --graph 0 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_traffic_$c.log 2>&1
  rc=$?; echo "$c rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R && python3 tools/pmc_step.py gpurun_out/pmc_traffic_FETCH_SIZE gpurun_out/pmc_traffic_WRITE_SIZE > gpurun_out/pmc_traffic.json; echo parsed rc=$?
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/pmc_traffic.json"))
for k, e in sorted(d.items(), key=lambda kv: -(kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0))):
    f, w = e.get("FETCH_SIZE", 0) * 1024 / 1e6, e.get("WRITE_SIZE", 0) * 1024 / 1e6
    if f + w > 1:
        print("%-50s fetch %8.1f MB  write %8.1f MB" % (k[:50], f, w))
PY
