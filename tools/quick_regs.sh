#!/bin/bash
# device-only compile of gm_api.hip and the register report of the default (32-byte block) search kernels -- no library is written
cd "$(dirname "$0")/../genmap_amd/csrc" || exit 1
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only --no-gpu-bundle-output -w -c gm_api.hip -o $T/dev.co "$@" || exit 1
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/dev.co | grep -E "\.name:|vgpr_count|sgpr_count|private_segment_fixed_size|vgpr_spill" | paste - - - - - | sed 's/  */ /g' | grep -E "search_kernel_w4ILi1" | sed 's/_ZN2gm16search_kernel_w4//; s/NS_10SearchArgsE//' | awk '{print $4, $2, $6, $8, $10}' 
rm -rf $T
