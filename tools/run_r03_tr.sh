# GPU box: kernel timeline of the TRAINER's replayed step (bench.py's `trainer` companion): what it adds to the bench step
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tr
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/tr -- python3 $R/bench.py --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --trainer-steps 64 > $R/gpurun_out/tr.log 2>&1 || { tail -5 $R/gpurun_out/tr.log; exit 1; }
python3 - $R/gpurun_out/tr/*/*kernel_trace.csv <<'PY'
import csv, sys
from collections import Counter
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('lnerf::', '').split('(')[0][:70]) for r in csv.DictReader(open(sys.argv[1]))))
starts = [i for i, r in enumerate(rows) if r[2].startswith('k_march_train<false')]
steps = [rows[a:b] for a, b in zip(starts[:-1], starts[1:])]
cnt = Counter(len(s) for s in steps)
print("kernels per step histogram:", sorted(cnt.items()))
# the trainer's steps: the most common count above the bench's 9
n = max((k for k in cnt if k > 9 and k < 40), key=lambda k: cnt[k])
good = [s for s in steps if len(s) == n][-20:]
spans = sorted(s[-1][1] - s[0][0] for s in good); med = spans[len(spans) // 2]
good = [s for s in good if s[-1][1] - s[0][0] <= 1.3 * med]
tot = 0
for j in range(n):
    d = sum(s[j][1] - s[j][0] for s in good) / len(good) / 1e3
    g = sum((s[j][0] - s[j - 1][1]) if j else 0 for s in good) / len(good) / 1e3
    tot += d
    print("%8.2f  (+%6.2f gap)  %s" % (d, g, good[0][j][2]))
print("kernels %.1f us; span %.1f us; step to step %.1f us" % (tot, sum(s[-1][1] - s[0][0] for s in good) / len(good) / 1e3,
      (good[-1][0][0] - good[0][0][0]) / (len(good) - 1) / 1e3 if len(good) > 1 else 0))
PY
