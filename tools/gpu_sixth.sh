#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "lines", rf.get("rank_lines"), "steps/kmer", rf.get("node_steps_per_kmer"))'
echo "== pytest gpu (parity, no cli)"; timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5
echo "== chr1 sweeps of GM_VERIFY_T"
for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do
  for T in 0 1 2 4; do echo -n "T=$T $cfg: "; GM_VERIFY_T=$T timeout 1200 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "$J"; done
done
