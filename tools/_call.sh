cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04
(timeout 1500 python -m pytest tests -m gpu -q --timeout 600 --durations=15 > gpurun_out/r04/pytest_gpu_full_pre.txt 2>&1; tail -25 gpurun_out/r04/pytest_gpu_full_pre.txt)
