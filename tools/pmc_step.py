#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc passes of an eager bench step per kernel.

Passes (each its own run; counters only ever combined with --kernel-trace):
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
            SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_sq_a \
            -- python3 bench.py --graph 0 --steps 6 --warmup 2 --no-cpu-baseline
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT \
            SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS ... -d gpurun_out/pmc_sq_b -- (same)
Usage: python tools/pmc_step.py gpurun_out/pmc_sq_a gpurun_out/pmc_sq_b > profiles/<name>.json
SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)."""
import csv
import glob
import json
import os
import re
import sys


def short(name):
    m = re.search(r"lnerf::(\w+)", name)
    if m:
        k = m.group(1)
        t = re.search(r"<([^>]*)>", name)
        return k + ("<" + t.group(1).replace(" ", "") + ">" if t else "")
    return name[:60]


def main():
    per = {}
    for d in sys.argv[1:]:
        for f in glob.glob(os.path.join(d, "*", "*counter_collection.csv")) + glob.glob(os.path.join(d, "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    out = {}
    for k, cs in per.items():
        if not k.startswith("k_"):
            continue
        e = {"launches": max(len(v) for v in cs.values())}
        for c, v in cs.items():
            e[c] = round(sum(v) / len(v), 1)
        wc = e.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS"):
                if c in e:
                    e["frac_" + c[3:].lower()] = round(e[c] / wc, 3)
        if e.get("SQ_WAVES") and e.get("SQ_INSTS_VALU"):
            e["valu_per_wave"] = round(e["SQ_INSTS_VALU"] / e["SQ_WAVES"], 1)
        out[k] = e
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
