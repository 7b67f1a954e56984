#!/usr/bin/env python3
"""Kernel timeline of the TRAINER's replayed step from a rocprofv3 kernel trace of bench.py's `trainer` companion
(tools/run_profile.sh trainer): per kernel of a step its mean duration and the idle gap before it, the span of a step and
the step-to-step time -- step-to-step minus span is what the GPU idles per step waiting for the host."""
import csv
import sys
from collections import Counter


def main():
    rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']),
                    r['Kernel_Name'].replace('void ', '').replace('lnerf::', '').split('(')[0][:70])
                   for r in csv.DictReader(open(sys.argv[1]))))
    starts = [i for i, r in enumerate(rows) if r[2].startswith('k_march_train<false')]
    steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
    cnt = Counter(len(s) for s in steps)
    print("kernels per step histogram:", sorted(cnt.items()))
    # the trainer's steps: the ones that contain the guidance launch (one HIP kernel since round 4; the torch form is
    # recognised by its kernel count: the most common one above the bench's 9)
    guided = [j for j, s in enumerate(steps) if any(r[2].startswith('k_synthetic_guidance') for r in s)]
    if guided:
        n = Counter(len(steps[j]) for j in guided).most_common(1)[0][0]
        idx = [j for j in guided if len(steps[j]) == n]
    else:
        n = max((k for k in cnt if 9 < k < 40), key=lambda k: cnt[k])
        idx = [j for j, s in enumerate(steps) if len(s) == n]
    good = [steps[j] for j in idx][-40:]
    spans = sorted(s[-1][1] - s[0][0] for s in good)
    med = spans[len(spans) // 2]
    good = [s for s in good if s[-1][1] - s[0][0] <= 1.3 * med]
    tot = 0
    for j in range(n):
        d = sum(s[j][1] - s[j][0] for s in good) / len(good) / 1e3
        g = sum((s[j][0] - s[j - 1][1]) if j else 0 for s in good) / len(good) / 1e3
        tot += d
        print("%8.2f  (+%6.2f gap)  %s" % (d, g, good[0][j][2]))
    # step to step over CONSECUTIVE trainer steps only
    s2s = [steps[b][0][0] - steps[a][0][0] for a, b in zip(idx[:-1], idx[1:]) if b == a + 1]
    s2s.sort()
    span = sum(s[-1][1] - s[0][0] for s in good) / len(good) / 1e3
    print("kernels %.1f us; span %.1f us; step to step median %.1f us (%d pairs) -> GPU idle per step %.1f us"
          % (tot, span, s2s[len(s2s) // 2] / 1e3 if s2s else 0, len(s2s), (s2s[len(s2s) // 2] / 1e3 - span) if s2s else 0))


if __name__ == "__main__":
    main()
