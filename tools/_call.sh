cd $GRAFT_REPO_ROOT
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 600 -k "gtest or baseline_settings_small or fixture or midsize or chr1_e0 or ecoli or interleaved or alignment or reproduces_reference or exclude or csv or sampled" 2>&1 | tail -8) > gpurun_out/c21_pytest.txt
(timeout 900 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,0,1.0 100,0,1.0 24,0,1.0 -- "" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c21_sweep_mirror.txt
(timeout 600 python tools/stats_run.py --workload grch38 --cfg 30,0 2>&1 | grep -v amdgpu.ids | cut -c1-1200) > gpurun_out/c21_stats.txt
