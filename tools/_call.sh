cd $GRAFT_REPO_ROOT
bash tools/rehearsal_diag.sh 1.0 0 p2p 420 "100,1:1" > gpurun_out/c3_rehearsal_full_p2p.txt 2>&1
bash tools/rehearsal_diag.sh 1.0 0 collective 420 "100,1:1" > gpurun_out/c3_rehearsal_full_collective.txt 2>&1
(timeout 900 python tools/sweep_tuning.py --workload grch38 --cfg 30,2,0.03 -- "" "oss_weights=30549" "oss_weights=34884" "oss_weights=21879" "oss_weights=30309" "oss_weights=30039" "oss_weights=34644" "oss_weights=39219" 2>&1 | grep -v amdgpu.ids) > gpurun_out/c3_sweep_ossw_e2.txt
