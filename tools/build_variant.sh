#!/bin/bash
# Experiment builds of the library (never the product library): tools/build_variant.sh NAME "-DFLAG=1 ..." [grid-only]
# -> latent-nerf-test_amd/lib/liblnerf_hip_NAME.so (build tag NAME), for tools/ab_bench.sh / ab_kernels.sh.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; EXTRA=${2:-}
C=$R/latent-nerf-test_amd/csrc
O=/tmp/lnerf_variant_$NAME
mkdir -p $O
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DLNERF_BUILD_TAG=\"$NAME\" $EXTRA"
for f in rays grid mlp mlp_bf16 composite optim bg mesh raster; do
  /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o $O/$f.o &
done
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -DLNERF_BUILD_TAG=\"$NAME\" -c $C/api.cc -o $O/api.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/latent-nerf-test_amd/lib/liblnerf_hip_$NAME.so $O/*.o
echo built $R/latent-nerf-test_amd/lib/liblnerf_hip_$NAME.so
