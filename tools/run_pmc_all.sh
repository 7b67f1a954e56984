# GPU box: every counter-derived number bench.py reports, collected on THIS build of the library (the JSON carries the
# build tag; bench.py nulls the fields when the tags differ).  One rocprofv3 --pmc pass per counter group, each only
# ever combined with --kernel-trace, the program directly after `--`.  Output: gpurun_out/<name>.json -> copy to
# profiles/pmc_latest.json (and profiles/<round>_pmc_all.json).
set -u
R=$GRAFT_REPO_ROOT
NAME=${1:-r03_pmc_all}
EXTRA=${2:-}   # e.g. "--gridtype blocked"
cd /tmp && export TMPDIR=/tmp
# (--refresh 0: the occupancy refresh launches the gather on 1 M unrelated points; it must stay out of the per-kernel means)
CMD="python3 $R/bench.py --graph 0 --steps 6 --warmup 2 --no-cpu-baseline --no-extras --refresh 0 $EXTRA"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
           "TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_ACCESSES_sum GRBM_GUI_ACTIVE" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CU_CYCLES SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU_MFMA_BF16 SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
  i=$((i+1))
  d=$R/gpurun_out/${NAME}_p$i
  rm -rf $d
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $d -- $CMD > $R/gpurun_out/${NAME}_p$i.log 2>&1
  rc=$?; echo "pass $i ($grp) rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R && python3 tools/pmc_all_parse.py gpurun_out/${NAME} > gpurun_out/${NAME}.json; echo "parse rc=$?"; head -c 1500 gpurun_out/${NAME}.json
