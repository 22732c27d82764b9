#!/bin/bash
# compile ONE instantiation of the search kernel for gfx950 (seconds instead of the library's two minutes) and print its resources;
# the assembly is left in /tmp/ok/one.s:   tools/one_kernel.sh [-DKWPP=1] [-DKENV='CountEnv<1,true>'] [-DKCOOP=true] [-DGM_...]
cd "$(dirname "$0")/.." && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -S -Igenmap_amd/csrc "$@" -o /tmp/ok/one.s /tmp/ok/one.hip 2>&1 | grep -E "error|warning: v" ; grep -E "\.vgpr_count|\.sgpr_count|\.private_segment_fixed_size|vgpr_spill_count|\.name:" /tmp/ok/one.s | paste - - - - - | sed 's/  */ /g'
