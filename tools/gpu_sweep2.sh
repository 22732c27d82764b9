#!/bin/bash
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print("val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer %.2f" % rf.get("node_steps_per_kmer"))'
echo "== E=2 K=30"; for L in 18 20 22 23 24 25; do echo -n "infix=$L: "; timeout 900 python bench.py --E 2 --steps 1 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=1 K=30"; for L in 25 26 27; do echo -n "infix=$L: "; timeout 900 python bench.py --E 1 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=1 K=100"; for L in 70 85 90 95; do echo -n "infix=$L: "; timeout 900 python bench.py --K 100 --E 1 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=0 K=100"; for L in 16 24 40 70; do echo -n "infix=$L: "; timeout 900 python bench.py --K 100 --E 0 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=3 K=50"; for L in 38 42 46; do echo -n "infix=$L: "; timeout 1200 python bench.py --K 50 --E 3 --scale 0.2 --steps 1 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=1 K=24"; for L in 19 20 21 22; do echo -n "infix=$L: "; timeout 900 python bench.py --K 24 --E 1 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
