#!/bin/bash
# needs variant libraries: hipcc ... -DGM_WAVES=5 -shared -o tools/exp/libw5.so gm_api.hip gm_build.hip (likewise w6); not kept in the tree
# occupancy experiment: forced waves/SIMD (spilling) vs the default 4
run() { python bench.py --no-cpu-baseline --no-counters --K $1 --E $2 --steps $3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g k-mers/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"; }
cp genmap_amd/lib/libgenmap_amd.so /tmp/lib_default.so
for v in "4 4 default" "5 2 w5" "6 1 w6" "5 2 default" "4 4 w5"; do set -- $v
  if [ "$3" = default ]; then cp /tmp/lib_default.so genmap_amd/lib/libgenmap_amd.so; else cp tools/exp/lib$3.so genmap_amd/lib/libgenmap_amd.so; fi
  export GM_BLOCKS_PER_CU=$1 GM_LDS_STACK=$2
  echo "== lib=$3 blocks/CU=$1 ldsStack=$2"
  echo -n "  K30E0: "; run 30 0 10; echo -n "  K30E1: "; run 30 1 3; echo -n "  K30E2: "; run 30 2 2
done
cp /tmp/lib_default.so genmap_amd/lib/libgenmap_amd.so
