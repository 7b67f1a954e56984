# GPU box: chunks of 64 lattice points per round of the march (4 / 8 / 16): kernel times + step rate + march parity tests
set -u
R=$GRAFT_REPO_ROOT
cd $R
L=latent-nerf-test_amd/lib
B="python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-extras --refresh 0"
for round in 1 2; do
for lib in liblnerf_hip.so liblnerf_hip_mc8.so liblnerf_hip_mc16.so; do
  LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 200 $B 2> gpurun_out/mc_err.log | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$lib', d['value'], d['ms_per_step'])" || { tail -5 gpurun_out/mc_err.log; exit 1; }
done
done
cd /tmp && export TMPDIR=/tmp
for lib in liblnerf_hip.so liblnerf_hip_mc8.so liblnerf_hip_mc16.so; do
rm -rf $R/gpurun_out/mc
LNERF_HIP_LIB=$R/$L/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/mc -- python3 $R/bench.py --steps 40 --warmup 6 --no-cpu-baseline --no-extras --refresh 0 > $R/gpurun_out/mc.log 2>&1 || exit 1
echo "== $lib"; python3 $R/tools/trace_timeline.py $R/gpurun_out/mc/*/*kernel_trace.csv | grep -i "march\|steps used"
done
cd $R
LNERF_HIP_LIB=$R/$L/liblnerf_hip_mc8.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -k "march or ray" > gpurun_out/mc_tests.log 2>&1; tail -2 gpurun_out/mc_tests.log
