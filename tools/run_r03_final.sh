# GPU box, round-3 evidence run: full -m gpu suite, smoke, default bench, kernel stats + timeline of the bench step,
# PMC passes (pmc_latest.json on THIS build), scatter PMC, force-dist line, short-run line.
set -u
R=$GRAFT_REPO_ROOT
NAME=${1:-r03_final}
cd $R
timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > gpurun_out/${NAME}_tests.log 2>&1
rc=$?; tail -3 gpurun_out/${NAME}_tests.log; [ $rc -ne 0 ] && exit $rc
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${NAME}_smoke.log 2>&1; tail -1 gpurun_out/${NAME}_smoke.log
bash tools/run_trace.sh ${NAME} > gpurun_out/${NAME}_trace.log 2>&1; tail -2 gpurun_out/${NAME}_trace.log
python3 tools/trace_timeline.py gpurun_out/${NAME}_kernel_trace.csv --json gpurun_out/${NAME}_step_timeline.json > gpurun_out/${NAME}_timeline.txt 2>&1; cat gpurun_out/${NAME}_timeline.txt
bash tools/run_pmc_all.sh ${NAME}_pmc_all > gpurun_out/${NAME}_pmc_all.log 2>&1; grep "rc=" gpurun_out/${NAME}_pmc_all.log
bash tools/run_pmc_all.sh ${NAME}_pmc_blocked "--gridtype blocked" > gpurun_out/${NAME}_pmc_blocked.log 2>&1; grep "rc=" gpurun_out/${NAME}_pmc_blocked.log
bash tools/run_pmc_scatter.sh ${NAME}_pmc_scatter > gpurun_out/${NAME}_pmc_scatter.log 2>&1; grep "rc=" gpurun_out/${NAME}_pmc_scatter.log
cp gpurun_out/${NAME}_pmc_all.json profiles/pmc_latest.json   # (so that the bench lines below carry the counter-derived fields)
timeout -k 10 400 python3 bench.py > gpurun_out/${NAME}_bench_default.json 2> gpurun_out/${NAME}_bench.err; rc=$?; cat gpurun_out/${NAME}_bench_default.json | head -c 4000; [ $rc -ne 0 ] && { tail -5 gpurun_out/${NAME}_bench.err; exit $rc; }
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/${NAME}_bench_steps20.json 2> gpurun_out/${NAME}_bench20.err; head -c 700 gpurun_out/${NAME}_bench_steps20.json; echo
timeout -k 10 300 python3 bench.py --force-dist --no-cpu-baseline > gpurun_out/${NAME}_bench_forcedist.json 2> gpurun_out/${NAME}_bench_forcedist.err; head -c 700 gpurun_out/${NAME}_bench_forcedist.json; echo
timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-extras --refresh 0 > gpurun_out/${NAME}_bench_norefresh.json 2> /dev/null; head -c 400 gpurun_out/${NAME}_bench_norefresh.json; echo
exit 0
