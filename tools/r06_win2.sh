#!/bin/bash
# 2-bit windows: parity on small texts first, then K=100 e=1 at 3.09 Gbp with win2 on / off
mkdir -p gpurun_out/r06w
GM_TEST_TIMEOUT=150 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "needle_windows_at_two_bits" > gpurun_out/r06w/parity.txt 2>&1
tail -5 gpurun_out/r06w/parity.txt
grep -q " passed" gpurun_out/r06w/parity.txt && ! grep -q "failed" gpurun_out/r06w/parity.txt || exit 1
timeout 900 python tools/sweep_tuning.py --workload grch38 --scale 1.0 --cfg 100,1,1.0 --reps 3 -- "win2=0" "win2=1" "win2=1,lds_stack=1" "win2=1,lds_stack=3" "win2=1,lds_stack=4" "win2=0" "win2=1" > gpurun_out/r06w/sweep_k100.txt 2>&1
tail -12 gpurun_out/r06w/sweep_k100.txt
