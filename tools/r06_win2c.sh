#!/bin/bash
# 2-bit windows at K < 64 (one chunk saved = one LDS stack level): the one-loop kernels of calls the split search does not take, and C5's locating kernel
mkdir -p gpurun_out/r06w
for r in 1 2; do for T in "win2=0" "win2=1"; do
  timeout 300 python bench.py --workload bacteria5 --steps 5 --no-cpu-baseline --no-counters --no-csv --extra-configs "" --tune "$T" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C5', d['config'].get('tune'), round(d['ms_per_step'],3), 'ms', '%.4g'%d['value'])"
done; done > gpurun_out/r06w/c5_win2.txt 2>&1
cat gpurun_out/r06w/c5_win2.txt
timeout 900 python tools/sweep_tuning.py --workload chr1 --cfg 30,1,1.0 30,2,0.3 36,2,0.2 50,1,1.0 24,1,1.0 --reps 2 -- "expand=0,win2=0" "expand=0,win2=1" "expand=0,win2=0" "expand=0,win2=1" > gpurun_out/r06w/sweep_short_windows.txt 2>&1
grep "^K=" gpurun_out/r06w/sweep_short_windows.txt
