#!/bin/bash
for cfg in "30 1" "30 2" "100 1"; do set -- $cfg; for pb in 0 1 2 3; do
  echo "== chr1 K=$1 E=$2 probation=$pb"
  GM_PROBATION=$pb python bench.py --no-cpu-baseline --no-counters --K $1 --E $2 --steps 3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g k-mers/s  %.3f ms/step' % (d['value'], d['ms_per_step']))"
done; done
