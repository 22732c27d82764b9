cd $GRAFT_REPO_ROOT
(timeout 600 python tools/sweep_tuning.py --workload bacteria5 --reps 2 --cfg 24,1,1.0 30,2,1.0 24,0,1.0 -- "" "STEP=1" "STEP=2" "STEP=4" "STEP=6" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c60_sweep_bact.txt
