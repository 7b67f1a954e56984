set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc_list.txt 2>&1
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"
B="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
rm -rf $R/gpurun_out/pmc_sq_a $R/gpurun_out/pmc_sq_b
timeout -k 10 240 rocprofv3 --pmc $A --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_a -- python3 $R/bench.py --graph 0 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_a.log 2>&1
rc=$?; echo "pass A rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
timeout -k 10 240 rocprofv3 --pmc $B --kernel-trace --output-format csv -d $R/gpurun_out/pmc_sq_b -- python3 $R/bench.py --graph 0 --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc_b.log 2>&1
rc=$?; echo "pass B rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
cd $R && python3 tools/pmc_step.py gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b > gpurun_out/pmc_step.json; echo parsed rc=$?
