cd $GRAFT_REPO_ROOT
(timeout 900 python tools/first_call_cost.py 2>&1 | grep -v amdgpu.ids) > gpurun_out/c40_first_call.txt
(timeout 1500 python tools/sweep_tuning.py --workload grch38 --reps 2 --cfg 30,1,0.2 30,2,0.03 100,1,0.5 -- "" "lds_stack=3" "lds_stack=2" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c40_sweep_lds.txt
