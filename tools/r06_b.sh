#!/bin/bash
O=gpurun_out/r06b; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_e2 -- python $GRAFT_REPO_ROOT/tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.03 --reps 2 -- "expand=1" > $GRAFT_REPO_ROOT/$O/prof_e2.txt 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_e2 -name "*kernel_stats*" | head -3
for f in $(find $O/prof_e2 -name "*kernel_stats.csv"); do head -12 $f | cut -c1-200; done
timeout 900 python tools/stats_run.py --workload grch38 --cfg 30,2 --frac 0.03 --settings "expand=0" "expand=1" "expand=1,sat_draw_w=1000000" > $O/stats_e2.txt 2>&1; tail -3 $O/stats_e2.txt | cut -c1-1500
timeout 900 python tools/stats_run.py --workload grch38 --cfg 30,1 --frac 0.1 --settings "expand=0" "expand=1,sat_draw_w=1000000" > $O/stats_e1.txt 2>&1; tail -2 $O/stats_e1.txt | cut -c1-1500
