#!/bin/bash
run() { python bench.py --no-cpu-baseline --no-counters --K $1 --E $2 --infix $3 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('K%dE%d infix %s: %.4g k-mers/s  %.3f ms/step' % (d['config']['K'], d['config']['E'], '$3', d['value'], d['ms_per_step']))"; }
for i in 28 27 26 25 24 23; do run 30 1 $i; done
for i in 26 25 24 23 22; do run 30 2 $i; done
for i in 90 85 80 75; do run 100 1 $i; done
