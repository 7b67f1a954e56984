# GPU box: SQ counters of the scatter kernels (two rocprofv3 --pmc passes over tools/bench_scatter.py)
set -u
R=$GRAFT_REPO_ROOT
NAME=${1:-r02_pmc_scatter}
cd /tmp && export TMPDIR=/tmp
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
rm -rf $R/gpurun_out/${NAME}_a $R/gpurun_out/${NAME}_b
timeout -k 10 240 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $R/gpurun_out/${NAME}_a -- python3 $R/tools/bench_scatter.py --rounds 4 --configs 3:0 > $R/gpurun_out/${NAME}_a.log 2>&1
rc=$?; echo "pass A rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 240 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $R/gpurun_out/${NAME}_b -- python3 $R/tools/bench_scatter.py --rounds 4 --configs 3:0 > $R/gpurun_out/${NAME}_b.log 2>&1
rc=$?; echo "pass B rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
cd $R && python3 tools/pmc_step.py gpurun_out/${NAME}_a gpurun_out/${NAME}_b > gpurun_out/${NAME}.json; echo parsed rc=$?
cat gpurun_out/${NAME}.json
