#!/bin/bash
O=gpurun_out/r06c; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=150
timeout 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_search" > $O/pytest_split.txt 2>&1; tail -5 $O/pytest_split.txt
grep -q "passed" $O/pytest_split.txt || exit 1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.03 30,1,0.1 100,1,0.3 --reps 2 -- "expand=0" "expand=1" "expand=1,sat_draw_w=4" "expand=1,sat_draw_w=1" "expand=1,fetch_batch=16" "expand=1,fetch_batch=2" > $GRAFT_REPO_ROOT/$O/ab.txt 2>&1
cd $GRAFT_REPO_ROOT
grep "K=" $O/ab.txt
for f in $(find $O/prof -name "*kernel_stats.csv"); do grep -E "expand_kernel|CountEnv|Scatter" $f | cut -c1-260; done
