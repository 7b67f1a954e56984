# GPU box, round 3 step N: scatter in level groups (bin(group) -> reduce(group)): records re-read while still in the Infinity Cache?
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid_encode_forward_backward or reproducible" > gpurun_out/r03n_tests.log 2>&1; tail -2 gpurun_out/r03n_tests.log
for i in 1 2 3; do
  for g in 1 2 4 8 16; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --refresh 0 --tune scatter_level_groups=$g > gpurun_out/r03n_g${g}_$i.json 2> gpurun_out/r03n_g${g}_$i.err || { tail -5 gpurun_out/r03n_g${g}_$i.err; exit 1; }
  done
done
python3 - <<'PY'
import json, glob, statistics
for g in (1, 2, 4, 8, 16):
    rows = [json.load(open(f)) for f in sorted(glob.glob("gpurun_out/r03n_g%d_*.json" % g))]
    print(json.dumps({"scatter_level_groups": g, "fps_median": round(statistics.median(r["value"] for r in rows), 1),
                      "scatter_ms_median": round(statistics.median(r["scatter"]["kernel_ms"] for r in rows), 4)}))
PY
exit 0
