#!/bin/bash
# VGPR/SGPR/LDS/scratch of every kernel in the library's gfx950 code object (unbundles the fat binary)
SO=${1:-genmap_amd/lib/libgenmap_amd.so}
B=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$B/clang-offload-bundler --type=o --input=$SO --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co 2>/dev/null || { objcopy -O binary --only-section=.hip_fatbin $SO $T/fat.bin && $B/clang-offload-bundler --type=o --input=$T/fat.bin --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co; }
$B/llvm-readelf --notes $T/dev.co | grep -E "\.name:|vgpr_count|sgpr_count|private_segment_fixed_size|group_segment_fixed_size|vgpr_spill" | paste - - - - - - | sed 's/  */ /g' | grep -E "search_kernel" | sed 's/gm::search_kernel//' | cut -c1-220
rm -rf $T
