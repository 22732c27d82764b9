#!/bin/bash
O=gpurun_out/r06i; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.1 --reps 1 -- "expand=0" "" "expand_overlap=1,blocks_per_cu=4,expand_occ=6" "expand_overlap=1,blocks_per_cu=4,expand_occ=3" "expand_overlap=1,blocks_per_cu=3,expand_occ=6" "expand_mb=24576" "expand_mb=24576,steal=8" "expand_mb=24576,steal=4" "expand_mb=24576,expand_overlap=1,blocks_per_cu=4,expand_occ=6" "expand_mb=49152,expand_overlap=1,blocks_per_cu=4,expand_occ=6" > $O/ab.txt 2>&1
grep "K=" $O/ab.txt
