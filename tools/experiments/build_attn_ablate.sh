#!/bin/bash
# builds tools/experiments/build/libattn4_abl{n}.so (run in the container; the .so files travel to the GPU box with the snapshot)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p tools/experiments/build
for a in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -DD3D_ATTN_ABL=$a -shared \
      -o tools/experiments/build/libattn4_abl$a.so dynam3d_amd/csrc/attn4_kernels.hip dynam3d_amd/csrc/d3d_error.cpp &
done
wait
ls -la tools/experiments/build
