#!/bin/bash
# 2-bit windows: other long-window settings with win2 off / on; batch knobs on top at K=100 e=1
mkdir -p gpurun_out/r06w
timeout 1200 python tools/sweep_tuning.py --workload grch38 --scale 1.0 --cfg 100,0,1.0 64,1,1.0 150,1,1.0 250,1,0.5 101,2,0.1 101,3,0.02 101,4,0.004 --reps 2 -- "win2=0" "win2=1" "win2=0" "win2=1" > gpurun_out/r06w/sweep_others.txt 2>&1
grep "^K=" gpurun_out/r06w/sweep_others.txt
timeout 600 python tools/sweep_tuning.py --workload grch38 --scale 0.5 --cfg 100,1,1.0 --reps 3 -- "win2=1" "win2=1,fetch_batch=32" "win2=1,fetch_batch=40" "win2=1,fetch_batch=56" "win2=1,steal=16" "win2=1,steal=48" "win2=1,verify_t_ext=8" "win2=1,pat_batch=4" "win2=1" > gpurun_out/r06w/sweep_k100_knobs.txt 2>&1
grep "^K=" gpurun_out/r06w/sweep_k100_knobs.txt
