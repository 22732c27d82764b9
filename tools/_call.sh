cd $GRAFT_REPO_ROOT
bash tools/profile_round.sh r03 > gpurun_out/c16_profile_round.log 2>&1
