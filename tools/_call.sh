cd $GRAFT_REPO_ROOT
(timeout 600 ./tools/gather_bench2 32 1 more 2>&1 | grep "footprint= 32768\|footprint=  8192\|device") > gpurun_out/c18_gather.txt
