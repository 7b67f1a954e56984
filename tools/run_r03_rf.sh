# GPU box: kernel timeline of ONE occupancy refresh (eager launches between graph replays): kernels vs gaps
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/rf
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/rf -- python3 $R/bench.py --steps 64 --warmup 6 --no-cpu-baseline --no-extras --repeats 1 > $R/gpurun_out/rf.log 2>&1 || { tail -5 $R/gpurun_out/rf.log; exit 1; }
python3 - $R/gpurun_out/rf/*/*kernel_trace.csv <<'PY'
import csv, sys
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('lnerf::', '').split('(')[0][:60]) for r in csv.DictReader(open(sys.argv[1]))))
# a refresh = the kernels between a k_scatter_reduce and the next k_march_train<false that are not the step's own
idx = [i for i, r in enumerate(rows) if r[2].startswith('k_occ_cell_points') or r[2].startswith('k_occ_sample') or 'occ_count' in r[2]]
# find refresh groups: sequences starting at first occ kernel after a reduce until the next march count pass
groups = []
i = 0
while i < len(rows):
    if 'occ' in rows[i][2] and (i == 0 or 'k_scatter_reduce' in rows[i - 1][2] or 'k_step_tail' in rows[i-1][2] or 'k_adam' in rows[i-1][2]):
        j = i
        while j < len(rows) and not rows[j][2].startswith('k_march_train<false'):
            j += 1
        groups.append(rows[i:j]); i = j
    else:
        i += 1
groups = [g for g in groups if len(g) > 3]
print(len(groups), 'refreshes; kernels each:', sorted(set(len(g) for g in groups)))
g = groups[len(groups) // 2]
t0 = g[0][0]; prev = None; ksum = 0
for s, e, n in g:
    gap = (s - prev) / 1e3 if prev else 0.0
    print("%8.2f us  (+%6.2f gap)  %s" % ((e - s) / 1e3, gap, n)); prev = e; ksum += (e - s) / 1e3
print("kernels %.1f us, span %.1f us" % (ksum, (g[-1][1] - t0) / 1e3))
PY
