#!/bin/bash
# round 6, first GPU run of the split search: parity on small texts, then A/B against the one-loop kernel on the 3.09 Gbp text
O=gpurun_out/r06a; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=150
timeout 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_search and (30-1 or 30-2)" > $O/pytest_split1.txt 2>&1; tail -15 $O/pytest_split1.txt
grep -q "passed" $O/pytest_split1.txt || exit 1
timeout 900 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.03 30,1,0.3 100,1,1.0 -- "expand=0" "expand=1" "expand=1,fetch_batch=1" "expand=1,fetch_batch=16" "expand=1,steal=0" "expand=1,sat_draw_w=1000000" > $O/ab_expand.txt 2>&1; tail -24 $O/ab_expand.txt
timeout 900 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split_search" > $O/pytest_split.txt 2>&1; tail -15 $O/pytest_split.txt
