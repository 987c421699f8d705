#!/bin/bash
# diagnostics build of the attention kernel with barrier stamps, run the report, restore the product build
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast -DD3D_FA_STAMP -c dynam3d_amd/csrc/attn2_kernels.hip -o dynam3d_amd/build/attn2_kernels.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dynam3d_amd/libdynam3d_hip.so dynam3d_amd/build/*.o
