#!/bin/bash
O=gpurun_out/r06h; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=150
timeout 500 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_search" > $O/pytest_split.txt 2>&1; tail -3 $O/pytest_split.txt
grep -q "passed" $O/pytest_split.txt || exit 1
timeout 1200 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.1 30,1,0.3 --reps 1 -- "expand=0" "" "fetch_batch=16" "fetch_batch=32" "steal=8" "steal=32" "expand_mb=6144" "expand_mb=24576" "expand_mb=40000" "probation=1" "verify_t_ext=1" "lds_stack=2" > $O/ab.txt 2>&1
grep "K=" $O/ab.txt
