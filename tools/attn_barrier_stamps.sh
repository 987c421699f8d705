#!/bin/bash
# Diagnostics build of the v2 attention kernel with barrier stamps (-DD3D_FA_STAMP) INTO the product library's place: run
#     tools/attn_barrier_stamps.sh && gpurun -- 'python tools/attn_barrier_stamps.py' ; tools/attn_barrier_stamps.sh --restore
# (the GPU box receives a snapshot of the tree, so the diagnostic library has to be in place when gpurun is called).
set -e
cd "$(dirname "$0")/.."
if [ "${1:-}" = "--restore" ]; then
    touch dynam3d_amd/csrc/attn2_kernels.hip
    python -m dynam3d_amd.build | tail -1
    exit 0
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -ffp-contract=fast -DD3D_FA_STAMP -c dynam3d_amd/csrc/attn2_kernels.hip -o dynam3d_amd/build/attn2_kernels.hip.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o dynam3d_amd/libdynam3d_hip.so dynam3d_amd/build/*.o
echo "diagnostic library in place (D3D_ATTN_V3=0 selects the stamped v2 kernel); restore with: tools/attn_barrier_stamps.sh --restore"
