# GPU box, round 3 step O: persistent reduce workgroups (sweep of the slot count), parity
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_render.py -x -q -m gpu -k "grid or scatter or fused or tail" > gpurun_out/r03o_tests.log 2>&1; rc=$?; tail -2 gpurun_out/r03o_tests.log; [ $rc -ne 0 ] && exit $rc
for i in 1 2 3; do
  for g in 65535 512 1024 768 256; do
    timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-extras --refresh 0 --tune scatter_reduce_wgs=$g > gpurun_out/r03o_g${g}_$i.json 2> gpurun_out/r03o_g${g}_$i.err || { tail -5 gpurun_out/r03o_g${g}_$i.err; exit 1; }
  done
done
python3 - <<'PY'
import json, glob, statistics
for g in (65535, 512, 1024, 768, 256):
    rows = [json.load(open(f)) for f in sorted(glob.glob("gpurun_out/r03o_g%d_*.json" % g))]
    print(json.dumps({"scatter_reduce_wgs": g, "fps_median": round(statistics.median(r["value"] for r in rows), 1),
                      "scatter_ms_median": round(statistics.median(r["scatter"]["kernel_ms"] for r in rows), 4)}))
PY
exit 0
