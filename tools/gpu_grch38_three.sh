#!/bin/bash
mkdir -p gpurun_out/r01h; O=gpurun_out/r01h
timeout 2400 python bench.py --workload grch38 --steps 3 > $O/bench_grch38_e0.json 2>/dev/null; tail -1 $O/bench_grch38_e0.json | cut -c60-140
timeout 2400 python bench.py --workload grch38 --E 1 --steps 2 --no-cpu-baseline > $O/bench_grch38_e1.json 2>/dev/null; tail -1 $O/bench_grch38_e1.json | cut -c60-140
timeout 3000 python bench.py --workload grch38 --K 100 --E 1 --steps 2 --no-cpu-baseline > $O/bench_grch38_k100e1.json 2>/dev/null; tail -1 $O/bench_grch38_k100e1.json | cut -c60-140
