#!/bin/bash
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print("val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer %.2f" % rf.get("node_steps_per_kmer"))'
echo "== E=0 infix sweep"
for L in 16 18 20 22 24 26 30; do echo -n "infix=$L: "; timeout 600 python bench.py --E 0 --steps 3 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=0 Q=13"
for L in 16 20; do echo -n "Q=13 infix=$L: "; GM_QTABLE=13 timeout 600 python bench.py --E 0 --steps 3 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=2 infix sweep"
for L in 24 26 27 28 30; do echo -n "infix=$L: "; timeout 900 python bench.py --E 2 --steps 1 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== K=100 E=1 infix sweep"
for L in 31 40 50 60 80; do echo -n "infix=$L: "; timeout 900 python bench.py --K 100 --E 1 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== grch38 E=0 infix sweep"
for L in 9 16 20 24; do echo -n "infix=$L: "; timeout 1200 python bench.py --workload grch38 --E 0 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
