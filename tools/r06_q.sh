#!/bin/bash
O=gpurun_out/r06q; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.1 30,1,0.3 --reps 1 -- "expand_mb=24576" "expand_mb=24576,steal=4" "expand_mb=24576,steal=12" "expand_mb=24576,fetch_batch=4" "expand_mb=24576,fetch_batch=16" "expand_mb=24576,sat_min_w=64" "expand_mb=24576,sat_min_w=16" "expand_mb=24576,probation=1" "expand_mb=24576,verify_cost=2" "expand_mb=24576,verify_cost=5" "expand_mb=24576,lds_stack=3" > $O/ab.txt 2>&1; grep "K=" $O/ab.txt
