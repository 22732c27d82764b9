#!/bin/bash
# the round's last GPU call: smoke, the whole -m gpu suite, then the measurement set of profiles/r06/final on the same tree
mkdir -p gpurun_out/r06t gpurun_out/r06final
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06t/smoke.txt 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/r06t/smoke.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/r06t/pytest_gpu_full.txt 2>&1; echo "suite rc=$?"; tail -12 gpurun_out/r06t/pytest_gpu_full.txt
grep -q " passed" gpurun_out/r06t/pytest_gpu_full.txt && ! grep -q "failed" gpurun_out/r06t/pytest_gpu_full.txt || exit 1
bash tools/profile_round6.sh bench stats trace pmc sq steps > gpurun_out/r06final/run.log 2>&1
tail -c 700 gpurun_out/r06final/bench_default.json
