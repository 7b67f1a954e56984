#!/bin/bash
# Re-creates tools/r02_tree (git-ignored): the tree of the last round-2 commit (0d6c8fe) with its own library built, for
# the same-box A/B of tools/run_r03_ab.sh / run_r03_s2.sh (`python3 tools/r02_tree/bench.py` loads ITS library).
# No GPU needed; run in the build container.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/tools/r02_tree
rm -rf $D && mkdir -p $D
git -C $R archive 0d6c8fe bench.py include latent-nerf-test_amd oracle | tar -x -C $D
(cd $D/latent-nerf-test_amd && python3 build.py)
echo "exported $(git -C $R rev-parse --short 0d6c8fe) to $D"
