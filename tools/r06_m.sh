#!/bin/bash
O=gpurun_out/r06m; mkdir -p $O; export TMPDIR=/tmp
export GM_TEST_TIMEOUT=200
timeout 400 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_correction_pass" > $O/pytest_piece.txt 2>&1; tail -3 $O/pytest_piece.txt
timeout 1200 python tools/sweep_tuning.py --workload grch38h --cfg 30,2,0.1 30,1,0.3 30,0,1.0 100,1,1.0 --reps 1 -- "expand=0" "" > $O/hard_text.txt 2>&1; grep -E "K=|index" $O/hard_text.txt
timeout 900 python tools/sweep_tuning.py --workload grch38 --sampling 10 --cfg 30,2,0.1 30,1,0.3 100,1,1.0 30,0,1.0 --reps 1 -- "" > $O/sampled_S10.txt 2>&1; grep -E "K=|index" $O/sampled_S10.txt
timeout 900 python tools/sweep_tuning.py --workload grch38 --sampling 0 --cfg 30,2,0.1 30,1,0.3 100,1,1.0 30,0,1.0 --reps 1 -- "" > $O/sampled_S0.txt 2>&1; grep -E "K=|index" $O/sampled_S0.txt
