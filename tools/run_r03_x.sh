# GPU box, run X: bin / reduce time of the coarse half (levels 0-7) and the fine half (8-15), from a two-group launch
set -u
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for G in 2 4; do
rm -rf $R/gpurun_out/x_$G
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/x_$G -- python3 $R/bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extras --refresh 0 --tune scatter_level_groups=$G > $R/gpurun_out/x_$G.log 2>&1 || { tail -5 $R/gpurun_out/x_$G.log; exit 1; }
python3 - $R/gpurun_out/x_$G/*/*kernel_trace.csv $G <<'PY'
import csv, sys
G = int(sys.argv[2])
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'k_scatter_bin' in r['Kernel_Name'] or 'k_scatter_reduce' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-2 * G * 20:]
for k in range(2 * G):
    d = sorted((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows[k::2 * G])
    print("G=%d launch %d %-18s median %.1f us" % (G, k, rows[k]['Kernel_Name'].split('<')[0].split('::')[-1], d[len(d) // 2]))
PY
done
