cd $GRAFT_REPO_ROOT
bash tools/rehearsal_diag.sh 1.0 0 p2p 420 "100,1:1" > gpurun_out/c39_rehearsal_p2p.txt 2>&1
