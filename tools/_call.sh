cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 900 -k "exclude or five_bacteria or fixture or csv or cli or alignment or sampled or locations or extremes" 2>&1 | tail -5) > gpurun_out/c57_pytest.txt
