#!/bin/bash
# 5 waves/SIMD after the VGPR diet (94 VGPRs): LDS per block must drop to 32 KB (3 stack levels) or below
for v in "4 4" "5 3" "5 2" "6 1"; do set -- $v
  export GM_BLOCKS_PER_CU=$1 GM_LDS_STACK=$2
  echo "== blocks/CU=$1 ldsStack=$2"; bash tools/gpu_abc.sh
done
