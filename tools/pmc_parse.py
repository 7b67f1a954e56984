#!/usr/bin/env python3
"""Parse rocprofv3 --pmc passes of `tools/microbench.py gather` into profiles/pmc_gather_latest.json.

Passes (each its own run, counters never combined with tracing domains other than --kernel-trace):
  rocprofv3 --pmc FETCH_SIZE                --kernel-trace --output-format csv -d gpurun_out/pmc_FETCH_SIZE -- python3 tools/microbench.py gather
  rocprofv3 --pmc WRITE_SIZE                --kernel-trace --output-format csv -d gpurun_out/pmc_WRITE_SIZE -- python3 tools/microbench.py gather
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum  --kernel-trace --output-format csv -d gpurun_out/pmc_TCC_HIT_sum_TCC_MISS_sum -- ...
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md §HBM).  gfx950 note from that guide: FETCH_SIZE reads
half the bytes of a WIDE coalesced stream; this kernel's reads are 4/8-byte gathers, an access width the guide
calls uncalibrated, so the raw value is reported (WRITE_SIZE matches the kernel's output bytes exactly)."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out")
out = {}
for d, names in (("pmc_FETCH_SIZE", ["FETCH_SIZE"]), ("pmc_WRITE_SIZE", ["WRITE_SIZE"]),
                 ("pmc_TCC_HIT_sum_TCC_MISS_sum", ["TCC_HIT_sum", "TCC_MISS_sum"])):
    files = glob.glob(os.path.join(src, d, "*", "*counter_collection.csv"))
    if not files:
        continue
    for r in csv.DictReader(open(files[0])):
        if "k_grid_forward" not in r["Kernel_Name"]:
            continue
        if int(r["Grid_Size"]) < 4 * 1024 * 1024:  # the XCD-pinned map (variant 1) launches a 1-D 1 Mi-thread grid: skip
            continue
        tt = "f32" if "k_grid_forward<float," in r["Kernel_Name"] else "bf16"
        to = "f32" if "float>(" in r["Kernel_Name"] else "bf16"
        out.setdefault("%s_table_%s_out" % (tt, to), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {"kernel": "k_grid_forward (variant 0)", "samples_per_launch": 430440, "source": "rocprofv3 --pmc, tools/pmc_parse.py",
       "variants": {}}
for k, v in out.items():
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    e = {"launches": len(next(iter(v.values()))), "FETCH_SIZE_KiB": mean.get("FETCH_SIZE"), "WRITE_SIZE_KiB": mean.get("WRITE_SIZE"),
         "TCC_HIT_sum": mean.get("TCC_HIT_sum"), "TCC_MISS_sum": mean.get("TCC_MISS_sum")}
    if e["FETCH_SIZE_KiB"] is not None and e["WRITE_SIZE_KiB"] is not None:
        e["hbm_bytes_per_launch"] = (e["FETCH_SIZE_KiB"] + e["WRITE_SIZE_KiB"]) * 1024
    if e["TCC_HIT_sum"]:
        e["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
    res["variants"][k] = e
path = os.path.join(ROOT, "profiles", "pmc_gather_latest.json")
json.dump(res, open(path, "w"), indent=1)
print(json.dumps(res, indent=1))
