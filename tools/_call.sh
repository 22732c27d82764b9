cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q --timeout 900 -k "sixteen or baseline_settings_small or fixture or chr1 or ecoli or gtest_matrix or shards or range or interleaved or exclude" 2>&1 | tail -5) > gpurun_out/c32_pytest.txt
(timeout 1500 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,1,0.2 30,2,0.03 100,1,0.5 -- "jump_filter=0" "" "probation=0" "probation=1" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c32_sweep.txt
