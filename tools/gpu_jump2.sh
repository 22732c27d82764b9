#!/bin/bash
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print("val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer", rf.get("node_steps_per_kmer"))'
echo "== pytest gpu parity"; timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for Q in 13 15; do for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do echo -n "Q=$Q $cfg: "; GM_QTABLE=$Q timeout 1200 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "$J"; done; done
echo "== grch38"; for Q in 13 15 16; do for cfg in "--E 0 --steps 2" "--E 2 --steps 1 --warmup 0"; do echo -n "Q=$Q $cfg: "; GM_QTABLE=$Q timeout 2400 python bench.py --workload grch38 $cfg --no-cpu-baseline --no-counters 2>/dev/null | python -c "$J"; done; done
