#!/bin/bash
# Diagnostic build of the library with phase stamps in the binning pass (-DLNERF_STAMPS); never the product library.
set -eu
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/latent-nerf-test_amd/csrc
O=$R/latent-nerf-test_amd/lib/stamps
mkdir -p $O
FLAGS="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -DLNERF_STAMPS -DLNERF_BUILD_TAG=\"stamps\""
for f in rays grid_gather grid_bin grid mlp mlp_bf16 composite optim bg mesh raster guidance; do
  /opt/rocm/bin/hipcc $FLAGS -c $C/$f.hip -o $O/$f.o &
done
g++ -O2 -std=c++17 -fPIC -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -DLNERF_BUILD_TAG=\"stamps\" -c $C/api.cc -o $O/api.o 2>/dev/null || /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -DLNERF_BUILD_TAG=\"stamps\" -c $C/api.cc -o $O/api.o
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/latent-nerf-test_amd/lib/liblnerf_hip_stamps.so $O/*.o
echo built $R/latent-nerf-test_amd/lib/liblnerf_hip_stamps.so
