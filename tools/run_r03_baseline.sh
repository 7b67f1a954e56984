# GPU box, round-3 baseline: LDS atomic microbenchmark, per-level scatter cost, binning-pass phase stamps, default bench
set -u
R=$GRAFT_REPO_ROOT
cd $R
timeout -k 10 120 tools/bin/lds_atomics > gpurun_out/r03_lds_atomics.txt 2>&1; cat gpurun_out/r03_lds_atomics.txt
LNERF_HIP_LIB=latent-nerf-test_amd/lib/liblnerf_hip_stamps.so timeout -k 10 200 python3 tools/bin_stamps.py > gpurun_out/r03_bin_stamps.json 2> gpurun_out/r03_bin_stamps.err || tail -5 gpurun_out/r03_bin_stamps.err
bash tools/run_levels_trace.sh > gpurun_out/r03_levels.txt 2>&1; cat gpurun_out/r03_levels.txt
timeout -k 10 300 python3 bench.py > gpurun_out/r03_base_bench.json 2> gpurun_out/r03_base_bench.err; tail -c 3000 gpurun_out/r03_base_bench.json
