#!/bin/bash
for t in 1 2 4 8; do for c in 2 3 5; do
  echo "== chr1 E0 T=$t cost=$c"
  GM_VERIFY_T=$t GM_VERIFY_COST=$c python bench.py --no-cpu-baseline --no-counters --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g k-mers/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
done; done
