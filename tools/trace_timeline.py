#!/usr/bin/env python3
"""Per-step timeline from a rocprofv3 kernel trace (tools/run_profile.sh timeline): finds the steady-state steps (delimited by the
ray-generation kernel (or the march count pass)), and prints for each kernel of a step its mean duration and the mean idle gap BEFORE it.

    python tools/trace_timeline.py gpurun_out/<name>_kernel_trace.csv [--json out.json]
"""
import csv
import json
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void ", "").replace("lnerf::", "")
    return n.split("(")[0][:70]


def main():
    rows = []
    for r in csv.DictReader(open(sys.argv[1])):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    # a step starts with ray generation, or (camera form: rays generated inside the march) with the march's count pass
    starts = [i for i, r in enumerate(rows) if r[2].startswith("k_get_rays")]
    if len(starts) < 4:
        starts = [i for i, r in enumerate(rows) if r[2].startswith("k_march_train<false")]
    steps = []
    for a, b in zip(starts[:-1], starts[1:]):
        steps.append(rows[a:b])
    # steady state: the most common kernel count per step, last 20 of those
    cnt = defaultdict(int)
    for s in steps:
        cnt[len(s)] += 1
    n = max(cnt, key=cnt.get)
    good = [s for s in steps if len(s) == n][-20:]
    # an eager probe step (bench.py --probe-every) has host-side gaps between its launches: keep the replayed steps
    spans = sorted(s[-1][1] - s[0][0] for s in good)
    med = spans[len(spans) // 2]
    good = [s for s in good if s[-1][1] - s[0][0] <= 1.2 * med]
    out = []
    tot_k = tot_g = 0.0
    for j in range(n):
        dur = sum(s[j][1] - s[j][0] for s in good) / len(good) / 1e3
        gap = sum((s[j][0] - s[j - 1][1]) if j else 0 for s in good) / len(good) / 1e3
        out.append({"kernel": good[0][j][2], "us": round(dur, 2), "gap_before_us": round(gap, 2)})
        tot_k += dur
        tot_g += gap
    span = sum(s[-1][1] - s[0][0] for s in good) / len(good) / 1e3
    res = {"steps_used": len(good), "kernels_per_step": n, "kernel_us": round(tot_k, 1), "gaps_us": round(tot_g, 1),
           "span_us": round(span, 1), "timeline": out}
    if "--json" in sys.argv:
        json.dump(res, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("steps used %d, kernels/step %d: kernels %.1f us + gaps %.1f us = span %.1f us" % (len(good), n, tot_k, tot_g, span))
    for o in out:
        print("%8.2f  (+%6.2f gap)  %s" % (o["us"], o["gap_before_us"], o["kernel"]))


if __name__ == "__main__":
    main()
