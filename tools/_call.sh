cd $GRAFT_REPO_ROOT
(timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_cli_end_to_end.py -x -q --timeout 400 -k "gtest or baseline_settings_small or fixture or exclude or csv or bacteria or sampled or locat or range_shares" 2>&1 | tail -25) > gpurun_out/c5_pytest.txt
(timeout 600 python tools/sweep_tuning.py --workload grch38 --reps 1 --cfg 30,1,0.1 30,2,0.03 100,1,0.5 -- "" "jump=0" "oss_weights=30549" "oss_weights=30549,jump=0" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c5_sweep_jump.txt
(timeout 400 python bench.py --workload bacteria5 --steps 2 --warmup 1 2>&1 | grep -v "amdgpu.ids" | tail -5) > gpurun_out/c5_bench_c5.txt
(timeout 300 python tools/sampling_cost.py 2>&1 | grep -v "amdgpu.ids" | tail -12) > gpurun_out/c5_sampling_cost.txt
