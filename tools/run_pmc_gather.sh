# rocprofv3 --pmc passes of the gather (GPU box).  One counter group per run, only ever combined with --kernel-trace.
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=pmc_$(echo $grp | tr ' ' '_')
  rm -rf $R/gpurun_out/$name
  timeout -k 10 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$name -- python3 $R/tools/microbench.py gather > $R/gpurun_out/$name.log 2>&1
  rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
done
cd $R && python3 tools/pmc_parse.py gpurun_out > gpurun_out/pmc_gather.json; echo "parse rc=$?"
