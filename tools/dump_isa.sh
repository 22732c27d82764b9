#!/bin/bash
# disassemble one kernel of the library's gfx950 code object: tools/dump_isa.sh <mangled-name-substring> [lib.so] > out.s
SO=${2:-genmap_amd/lib/libgenmap_amd.so}; B=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
objcopy -O binary --only-section=.hip_fatbin $SO $T/fat.bin && $B/clang-offload-bundler --type=o --input=$T/fat.bin --unbundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/dev.co
$B/llvm-objdump -d --no-show-raw-insn $T/dev.co | awk -v pat="$1" '$0 ~ pat && /^[0-9a-f]+ </{f=1} f{print} /s_endpgm/{if(f) exit}' | sed 's#//.*##'
rm -rf $T
