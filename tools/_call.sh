cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q --timeout 600 -k "gtest or baseline_settings_small or fixture or midsize or chr1 or ecoli or interleaved or exclude or csv or sampled or range_shares" 2>&1 | tail -8) > gpurun_out/c22_pytest.txt
(timeout 1200 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,0,1.0 30,1,0.2 30,2,0.03 100,1,0.5 -- "" "steal=0" "fetch_batch=16" "fetch_batch=8" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c22_sweep_ahead.txt
