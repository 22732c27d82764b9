#!/bin/bash
mkdir -p gpurun_out
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"])'
echo "== pytest gpu parity"; timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -5
echo "== chr1 all configs, LDS stack depth sweep"
for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do
  for L in 0 2 4 8; do echo -n "LDS_STACK=$L $cfg: "; GM_LDS_STACK=$L timeout 1200 python bench.py $cfg --no-cpu-baseline --no-counters 2>/dev/null | python -c "$J"; done
done
for pc in 3 4 6; do echo -n "perCU=$pc E=2: "; GM_BLOCKS_PER_CU=$pc timeout 900 python bench.py --E 2 --steps 1 --no-cpu-baseline --no-counters 2>/dev/null | python -c "$J"; done
