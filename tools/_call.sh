cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py -x -q --timeout 300 -k "sixteen or baseline_settings_small or gtest_matrix or fixture or chr1" 2>&1 | tail -6) > gpurun_out/c48_pytest.txt
(timeout 900 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,1,0.2 30,2,0.03 100,1,0.5 150,1,0.5 50,2,0.05 -- "self_hit=0" "" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c48_sweep.txt
