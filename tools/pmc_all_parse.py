#!/usr/bin/env python3
"""rocprofv3 --pmc passes of an eager bench step (tools/run_pmc_all.sh) -> one JSON: per kernel the mean counter
values per launch and its mean duration, plus the derived numbers bench.py reports:

  gather[<table>_table_<feat>_out].hbm_bytes_per_launch = (FETCH_SIZE + WRITE_SIZE) KiB x 1024   (raw: the gather's 4/8-byte
        accesses are an access width MI355X_MICROARCH.md calls uncalibrated; WRITE_SIZE matches the output bytes)
  gather[...].l1_accesses_per_clk_cu = TCP_TOTAL_CACHE_ACCESSES / (kernel cycles x 256 CUs)   (the L1's tag path takes one)
  mfma[fwd|bwd].busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES)   (matrix-pipe cycles / SIMD cycles the
        kernel's CUs were busy)

usage: pmc_all_parse.py gpurun_out/<name>   (reads <name>_p*/ and <name>_p1.log for the build tag)"""
import csv
import glob
import json
import re
import sys


def short(name):
    m = re.search(r"lnerf::(\w+)", name)
    if not m:
        return None
    t = re.search(r"<([^>]*)>", name)
    return m.group(1) + ("<" + t.group(1).replace(" ", "").replace("lnerf::", "") + ">" if t else "")


def main():
    base = sys.argv[1]
    per, dur = {}, {}
    for f in glob.glob(base + "_p*/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    for f in glob.glob(base + "_p*/*/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if k:
                dur.setdefault(k, []).append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    build = None
    for f in sorted(glob.glob(base + "_p*.log")):
        m = re.search(r'"build": "([^"]+)"', open(f).read())
        if m:
            build = m.group(1)
            break
    kernels = {}
    for k, cs in per.items():
        e = {c: sum(v) / len(v) for c, v in cs.items()}
        e["launches"] = max(len(v) for v in cs.values())
        if k in dur:   # (durations under counter collection: a few % longer than unprofiled)
            e["mean_ns_profiled"] = sum(dur[k]) / len(dur[k])
        kernels[k] = e
    out = {"build": build, "source": "rocprofv3 --pmc (tools/run_pmc_all.sh, tools/pmc_all_parse.py): bench.py --graph 0 "
                                     "--steps 6 --warmup 2", "kernels": kernels, "gather": {}, "mfma": {}}
    for k, e in kernels.items():
        if k.startswith("k_grid_forward"):
            tt = "bf16" if "<unsignedshort," in k else "f32"
            to = "bf16" if k.endswith(",unsignedshort>") else "f32"
            g = {}
            if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
                g["hbm_bytes_per_launch"] = (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024
                g["fetch_bytes"], g["write_bytes"] = e["FETCH_SIZE"] * 1024, e["WRITE_SIZE"] * 1024
            if e.get("TCC_HIT_sum"):
                g["l2_hit_rate"] = e["TCC_HIT_sum"] / (e["TCC_HIT_sum"] + e["TCC_MISS_sum"])
            if "TCP_TOTAL_CACHE_ACCESSES_sum" in e:
                g["l1_cache_accesses"] = e["TCP_TOTAL_CACHE_ACCESSES_sum"]
                g["l1_to_l2_read_requests"] = e.get("TCP_TCC_READ_REQ_sum")
                if "GRBM_GUI_ACTIVE" in e:   # summed over the 8 XCDs
                    cyc = e["GRBM_GUI_ACTIVE"] / 8.0
                    g["kernel_cycles"] = cyc
                    g["l1_accesses_per_clk_cu"] = e["TCP_TOTAL_CACHE_ACCESSES_sum"] / (cyc * 256.0)
            out["gather"]["%s_table_%s_out" % (tt, to)] = g
        if k.startswith("k_mlp_forward_bf16") or (k.startswith("k_mlp_backward_bf16") and "_sw" not in k):
            m = {c: e.get(c) for c in ("SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INSTS_VALU_MFMA_BF16",
                                       "SQ_BUSY_CU_CYCLES", "SQ_VALU_MFMA_COEXEC_CYCLES", "GRBM_GUI_ACTIVE", "SQ_WAVES")}
            if m["SQ_VALU_MFMA_BUSY_CYCLES"] and m["SQ_BUSY_CU_CYCLES"]:
                m["busy_frac"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * m["SQ_BUSY_CU_CYCLES"])
            if m["SQ_VALU_MFMA_BUSY_CYCLES"] and m["GRBM_GUI_ACTIVE"]:
                m["busy_frac_of_chip"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0)
            out["mfma"]["fwd" if "forward" in k else "bwd"] = m
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == "__main__":
    main()
