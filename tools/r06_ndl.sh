#!/bin/bash
# verification reads the needle from the LDS window of the item's root: parity on small texts first, then A/B at 3.09 Gbp
mkdir -p gpurun_out/r06w
GM_TEST_TIMEOUT=200 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "needle_windows or split_search or sampled_suffix or exclude_pseudo_and_locations or long_kmers or interleaved or correction or hung" > gpurun_out/r06w/ndl_parity.txt 2>&1
tail -4 gpurun_out/r06w/ndl_parity.txt
grep -q " passed" gpurun_out/r06w/ndl_parity.txt && ! grep -q "failed" gpurun_out/r06w/ndl_parity.txt || exit 1
timeout 1200 python tools/sweep_tuning.py --workload grch38 --scale 1.0 --cfg 100,1,1.0 150,1,1.0 64,1,0.5 101,2,0.1 50,2,0.1 30,2,0.1 --reps 2 -- "needle_lds=0" "needle_lds=1" "needle_lds=0" "needle_lds=1" > gpurun_out/r06w/sweep_needle_lds.txt 2>&1
grep "^K=" gpurun_out/r06w/sweep_needle_lds.txt
