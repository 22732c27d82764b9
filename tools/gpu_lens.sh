#!/bin/bash
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print("val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer %.2f" % rf.get("node_steps_per_kmer"), "items", rf.get("verify_items"))'
echo "== E=2 K=30 infix 24, block lengths"
for LENS in "6,6,6,6" "4,4,8,8" "5,3,8,8" "3,3,9,9" "2,2,10,10" "6,2,8,8" "7,1,8,8" "4,2,9,9" "8,8,4,4"; do echo -n "lens=$LENS: "; GM_OSS_LENGTHS=$LENS timeout 900 python bench.py --E 2 --steps 1 --infix 24 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=2 K=30 infix 26"
for LENS "in" "7,7,6,6"; do :; done
for LENS in "7,7,6,6" "4,4,9,9" "3,3,10,10" "5,3,9,9"; do echo -n "lens=$LENS: "; GM_OSS_LENGTHS=$LENS timeout 900 python bench.py --E 2 --steps 1 --infix 26 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=1 K=30 infix 26"
for LENS in "13,13" "12,14" "14,12" "10,16" "16,10"; do echo -n "lens=$LENS: "; GM_OSS_LENGTHS=$LENS timeout 900 python bench.py --E 1 --steps 2 --infix 26 --no-cpu-baseline 2>/dev/null | python -c "$J"; done
