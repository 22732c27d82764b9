#!/bin/bash
O=gpurun_out/r06l; mkdir -p $O; export TMPDIR=/tmp
timeout 1000 python -u -m pytest tests/test_gpu_parity.py -m gpu -x -q --durations=8 -k "eight_ranks or full_size_grch38" > $O/pytest_new.txt 2>&1; tail -25 $O/pytest_new.txt
