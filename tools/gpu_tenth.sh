#!/bin/bash
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer", rf.get("node_steps_per_kmer"))'
echo "== pytest gpu parity"; timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for Q in 0 12; do for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do echo -n "Q=$Q $cfg: "; GM_QTABLE=$Q timeout 1200 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "$J"; done; done
echo "== E=0 infix sweep (Q=12)"
for L in 7 9 11 12 13 14 16; do echo -n "infix=$L: "; timeout 600 python bench.py --E 0 --steps 3 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
echo "== E=1 infix sweep"
for L in 20 22 24 26 28; do echo -n "infix=$L: "; timeout 600 python bench.py --E 1 --steps 2 --infix $L --no-cpu-baseline 2>/dev/null | python -c "$J"; done
