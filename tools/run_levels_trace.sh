set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/lv_trace
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/lv_trace -- python3 $R/tools/microbench.py scatter_levels > $R/gpurun_out/lv.log 2>&1
rc=$?; echo "rc=$rc"; [ $rc -ne 0 ] && exit $rc
cd $R && python3 - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/lv_trace/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'k_scatter' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# per-level calls: 7 calls per level (1 warm + 6 timed), levels interleaved per round: order = warm: l0..l15, then 6 rounds
bins = [r for r in rows if 'k_scatter_bin' in r['Kernel_Name']]
reds = [r for r in rows if 'k_scatter_reduce' in r['Kernel_Name']]
print(len(bins), len(reds))
def dur(r): return (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1000.0
nl = 16
for l in range(nl):
    b = sorted(dur(bins[i]) for i in range(l, len(bins), nl))
    r = sorted(dur(reds[i]) for i in range(l, len(reds), nl))
    print("level %2d  bin %.1f us  reduce %.1f us  grid_red %s" % (l, b[len(b)//2], r[len(r)//2], reds[l].get('Grid_Size', reds[l].get('Grid_Size_X', '?'))))
PY
