cd $GRAFT_REPO_ROOT
(timeout 1500 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,0,1.0 50,0,1.0 100,0,1.0 -- "jump_filter=0" "" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c36_sweep.txt
(timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 900 -k "not more_than_2_to_32 and not two_ranks and not three_ranks" 2>&1 | tail -5) > gpurun_out/c36_pytest.txt
