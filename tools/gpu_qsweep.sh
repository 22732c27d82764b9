#!/bin/bash
# q-mer table length sweep for e = 0 (first OSS block = whole infix)
for wl in chr1 grch38; do
  for q in 12 13 14 15; do
    echo "== $wl q=$q"
    GM_QTABLE=$q python bench.py --no-cpu-baseline --no-counters --workload $wl --steps 10 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.4g k-mers/s  %.3f ms/step kernel %.3f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
  done
done
for q in 12 14; do echo "== chr1 K100 E1 q=$q"; GM_QTABLE=$q python bench.py --no-cpu-baseline --no-counters --K 100 --E 1 --steps 5 2>&1 | tail -1 | cut -c60-140; done
