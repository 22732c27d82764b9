#!/bin/bash
O=gpurun_out/r06r; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python tools/sweep_tuning.py --workload grch38 --cfg 100,1,1.0 64,1,0.5 100,2,0.1 --reps 1 -- "expand=0" "expand=1" "expand=1,expand_two_pass=0" "expand=1,fetch_batch=32" "expand=1,steal=0" > $O/ab.txt 2>&1; grep "K=" $O/ab.txt
