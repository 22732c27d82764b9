#!/bin/bash
export TMPDIR=/tmp
J='import sys,json; r=json.loads(sys.stdin.read()); rf=r["roofline"]; print(r["config"]["workload"][:40], "val %.4g" % r["value"], "kernel_ms %.2f" % rf["kernel_ms"])'
echo "== pytest gpu parity"; timeout 2400 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
for cfg in "--E 0 --steps 3" "--E 1 --steps 2" "--E 2 --steps 1" "--K 100 --E 1 --steps 2"; do echo -n "$cfg: "; timeout 1200 python bench.py $cfg --no-cpu-baseline --no-counters 2>/dev/null | python -c "$J"; done
