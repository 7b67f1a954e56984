#!/bin/bash
# GPU box: the gather at REDUCED occupancy -- what a kernel fused with the MLP forward (3 waves per SIMD, DESIGN.md section 4
# H7) would leave it.  gather_lds_pad reserves unused LDS per 256-thread workgroup: 0 = 8 waves per SIMD (the kernel's own
# limit), 32768 -> 5 (the 160 KiB of a CU hold five 32 KiB reservations ...), 40960 -> 4, 53248 -> 3, 65536 -> 2.
# One JSON line per setting: gather kernel time from the live HIP-event probes of the bench step.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
for pad in 0 32768 40960 53248 65536; do
  timeout -k 10 200 python3 $R/bench.py --steps 64 --warmup 8 --repeats 1 --refresh 0 --no-cpu-baseline --no-extras --tune gather_lds_pad=$pad 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'gather_lds_pad': $pad, 'gather_kernel_ms': round(d['roofline']['kernel_ms'],5), 'frac_of_hbm_peak': round(d['roofline']['frac'],4), 'mlp_fwd_ms': round(d['mfma']['fwd_ms'],5), 'ms_per_step': round(d['ms_per_step'],5)}))" || exit 1
done
