#!/usr/bin/env python3
"""Per-kernel resource usage of the HIP sources (no GPU needed): VGPRs, AGPRs, scratch bytes per lane, LDS, occupancy,
from hipcc's -Rpass-analysis=kernel-resource-usage remarks.

    python tools/resource_usage.py [file.hip ...] [-D...]        default: every csrc/*.hip

A kernel that shows ScratchSize > 0 spills: in the streaming loops of this path that is poison (scratch traffic shares the
vector-memory queue with the loads the loop is waiting for)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "latent-nerf-test_amd", "csrc")
KEYS = ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]")


def demangle(names):
    p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return p.stdout.splitlines() if p.returncode == 0 else names


def usage(files=None, defs=()):
    """[{file, name (demangled, short), VGPRs, AGPRs, scratch, occupancy, lds}] for the kernels of `files`."""
    files = files or sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    rows = []
    for f in files:
        path = f if os.path.exists(f) else os.path.join(CSRC, f)
        cmd = ["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-ffp-contract=off",
               '-DLNERF_BUILD_TAG="ru"', "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", "/dev/null"] + list(defs)
        err = subprocess.run(cmd, capture_output=True, text=True).stderr
        cur = None
        for line in err.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"file": os.path.basename(path), "name": m.group(1)}
                rows.append(cur)
                continue
            for key in KEYS:
                m = re.search(re.escape(key) + r": (\d+)", line)
                if m and cur is not None and key not in cur:
                    cur[key] = int(m.group(1))
    for r, n in zip(rows, demangle([r["name"] for r in rows])):
        r["kernel"] = re.sub(r"\(.*", "", n.replace("lnerf::", "").replace("void ", ""))
        r["scratch"] = r.get("ScratchSize [bytes/lane]", 0)
    return rows


def main():
    args = sys.argv[1:]
    rows = usage([a for a in args if not a.startswith("-D")], [a for a in args if a.startswith("-D")])
    print("%-14s %5s %5s %7s %4s %7s  %s" % ("file", "VGPR", "AGPR", "scratch", "occ", "LDS", "kernel"))
    for r in rows:
        print("%-14s %5d %5d %7d %4d %7d  %s%s" % (r["file"], r.get("VGPRs", 0), r.get("AGPRs", 0), r["scratch"],
                                                  r.get("Occupancy [waves/SIMD]", 0), r.get("LDS Size [bytes/block]", 0),
                                                  r["kernel"], "   <-- SPILLS" if r["scratch"] else ""))
    print("%d kernels, %d with scratch" % (len(rows), sum(r["scratch"] > 0 for r in rows)))


if __name__ == "__main__":
    main()
