#!/bin/bash
O=gpurun_out/r06n; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=200
timeout 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_search" > $O/pytest_split.txt 2>&1; tail -3 $O/pytest_split.txt
grep -q "passed" $O/pytest_split.txt || exit 1
timeout 1200 python tools/sweep_tuning.py --workload grch38h --cfg 30,2,0.1 30,1,0.3 --reps 1 -- "expand=0" "expand_two_pass=0" "expand_two_pass=1" > $O/hard_text.txt 2>&1; grep -E "K=" $O/hard_text.txt
timeout 1200 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.1 30,1,0.3 --reps 1 -- "expand=0" "expand_two_pass=0" "expand_two_pass=1" > $O/easy_text.txt 2>&1; grep -E "K=" $O/easy_text.txt
