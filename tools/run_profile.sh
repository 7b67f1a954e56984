#!/bin/bash
# GPU box: ONE parametrised profiling driver (replaces the per-experiment run_r0x_*.sh wrappers).
#
#   bash tools/run_profile.sh <mode> <name> [extra bench.py flags ...]
#
#   stats      rocprofv3 --kernel-trace --stats of a short bench run       -> gpurun_out/<name>_kernel_stats.csv
#   timeline   kernel trace of replayed steps -> per-kernel mean + gaps     -> gpurun_out/<name>_step_timeline.json
#   trainer    kernel + memcpy trace of bench.py's `trainer` companion      -> gpurun_out/<name>_trainer_timeline.txt
#   pmc        every counter group bench.py reports (tools/run_pmc_all.sh)  -> gpurun_out/<name>.json
#   ab         per-kernel times for several library builds: extra args = lib1.so lib2.so ...
#   bench      plain bench line                                             -> gpurun_out/<name>_bench.json
#   evidence   -m gpu suite + default bench line + stats + timeline (a round's evidence run)
#
# rocprofv3 is only ever given the program itself after `--` (python3 ...), and --pmc is never combined with anything
# but --kernel-trace (see the pool's rules).  Every GPU step is bounded by `timeout -k 10`.
set -u
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
MODE=${1:?mode}; NAME=${2:?name}; shift 2
OUT=$R/gpurun_out
mkdir -p $OUT
BENCH="python3 $R/bench.py --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp

stats() {
  rm -rf $OUT/${NAME}_prof
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${NAME}_prof -- $BENCH --steps 30 --warmup 5 "$@" > $OUT/${NAME}_prof.log 2>&1
  rc=$?; echo "rocprof stats rc=$rc"; [ $rc -ne 0 ] && { tail -5 $OUT/${NAME}_prof.log; return $rc; }
  cp $OUT/${NAME}_prof/*/*kernel_stats.csv $OUT/${NAME}_kernel_stats.csv
  head -16 $OUT/${NAME}_kernel_stats.csv | cut -c1-160
}

timeline() {
  rm -rf $OUT/${NAME}_trace
  timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/${NAME}_trace -- $BENCH --steps 40 --warmup 5 --repeats 1 --refresh 0 "$@" > $OUT/${NAME}_trace.log 2>&1
  rc=$?; echo "rocprof trace rc=$rc"; [ $rc -ne 0 ] && { tail -5 $OUT/${NAME}_trace.log; return $rc; }
  python3 $R/tools/trace_timeline.py $OUT/${NAME}_trace/*/*kernel_trace.csv --json $OUT/${NAME}_step_timeline.json | tee $OUT/${NAME}_step_timeline.txt
}

trainer() {
  rm -rf $OUT/${NAME}_tr
  timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $OUT/${NAME}_tr -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --trainer-steps 96 "$@" > $OUT/${NAME}_tr.log 2>&1
  rc=$?; echo "rocprof trainer rc=$rc"; [ $rc -ne 0 ] && { tail -5 $OUT/${NAME}_tr.log; return $rc; }
  python3 $R/tools/trainer_timeline.py $OUT/${NAME}_tr/*/*kernel_trace.csv | tee $OUT/${NAME}_trainer_timeline.txt
}

case $MODE in
  stats) stats "$@" ;;
  timeline) timeline "$@" ;;
  trainer) trainer "$@" ;;
  pmc) bash $R/tools/run_pmc_all.sh $NAME "$*" ;;
  ab) bash $R/tools/ab_kernels.sh "$@" ;;
  bench)
    timeout -k 10 600 python3 $R/bench.py "$@" > $OUT/${NAME}_bench.json 2> $OUT/${NAME}_bench.err
    rc=$?; echo "bench rc=$rc"; [ $rc -ne 0 ] && tail -5 $OUT/${NAME}_bench.err; head -c 600 $OUT/${NAME}_bench.json; exit $rc ;;
  evidence)
    cd $R
    timeout -k 10 1100 python3 -m pytest tests -x -q -m gpu > $OUT/${NAME}_tests.log 2>&1
    rc=$?; tail -3 $OUT/${NAME}_tests.log; [ $rc -ne 0 ] && exit $rc
    timeout -k 10 400 python3 bench.py --breakdown > $OUT/${NAME}_bench_default.json 2> $OUT/${NAME}_bench.err
    rc=$?; head -c 400 $OUT/${NAME}_bench_default.json; [ $rc -ne 0 ] && { tail -5 $OUT/${NAME}_bench.err; exit $rc; }
    cd /tmp
    stats && timeline ;;
  *) echo "unknown mode $MODE"; exit 2 ;;
esac
