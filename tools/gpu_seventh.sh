#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"], "steps/kmer %.1f" % rf.get("node_steps_per_kmer"))'
echo "== shards test"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k shards 2>&1 | tail -30
echo "== chr1 sweeps of GM_VERIFY_T x COST"
for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do
  for TC in "1 0" "1 3" "4 2" "4 3" "4 5" "4 8"; do set -- $TC; echo -n "T=$1 C=$2 $cfg: "; GM_VERIFY_T=$1 GM_VERIFY_COST=$2 timeout 1200 python bench.py $cfg --no-cpu-baseline 2>/dev/null | python -c "$J"; done
done
