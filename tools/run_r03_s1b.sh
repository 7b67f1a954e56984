# GPU box: same-box A/B of stage 1 (last slice finishes the bucket) against the previous commit's library
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2 3; do
for lib in liblnerf_hip_prev.so liblnerf_hip.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/s1_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'], d['scatter']['kernel_ms'])" || { tail -5 gpurun_out/s1_err.log; exit 1; }
done
done
cd /tmp && export TMPDIR=/tmp
for lib in liblnerf_hip_prev.so liblnerf_hip.so; do
rm -rf $R/gpurun_out/s1b
LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/s1b -- python3 $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-extras --refresh 0 > $R/gpurun_out/s1b.log 2>&1 || exit 1
echo "== $lib"; python3 $R/tools/trace_timeline.py $R/gpurun_out/s1b/*/*kernel_trace.csv | grep -i "scatter\|tail\|steps used"
done
