#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-counters --K $1 --E $2 --steps $3 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K%dE%d %.4g k-mers/s  %.3f ms/step' % (d['config']['K'], d['config']['E'], d['value'], d['ms_per_step']))"; }
for b in 1 4 8 16 32; do echo "== fetch batch $b"; export GM_FETCH_BATCH=$b; run 30 0 10; run 30 1 3; run 30 2 2; run 100 1 3; done
