# Where does the gather wait?  TA/TCP counters (one group per run, only with --kernel-trace).
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
# (two counters per block and run: five TA counters at once exceed the hardware's slots and abort the profiler)
for grp in "TA_BUSY_avr GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1)); name=pmc_ta_$i
  rm -rf $R/gpurun_out/$name
  timeout -k 10 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $R/gpurun_out/$name -- python3 $R/tools/microbench.py gather > $R/gpurun_out/$name.log 2>&1
  rc=$?; echo "$name rc=$rc"; if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then exit $rc; fi
  if [ $rc -ne 0 ]; then grep -i "error code" $R/gpurun_out/$name.log | head -2; fi
done
cd $R && python3 - <<'PY'
import csv, glob, json
out = {}
for d in ("pmc_ta_1", "pmc_ta_2", "pmc_ta_3", "pmc_ta_4"):
    for f in glob.glob("gpurun_out/%s/*/*counter_collection.csv" % d):
        for r in csv.DictReader(open(f)):
            if "k_grid_forward<unsigned short, unsigned short>" not in r["Kernel_Name"] or int(r["Grid_Size"]) < 4 * 1024 * 1024:
                continue
            out.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in out.items()}
res["launches"] = max(len(v) for v in out.values()) if out else 0
json.dump(res, open("gpurun_out/pmc_gather_ta.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
