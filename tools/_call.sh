cd $GRAFT_REPO_ROOT
(timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "rehearsal or range_shares or interleaved" 2>&1 | tail -15) > gpurun_out/c2_pytest.txt
(timeout 300 python bench.py --workload chr1 --scale 0.05 --steps 3 --warmup 1 2>&1 | grep -v "amdgpu.ids" | tail -12) > gpurun_out/c2_bench_n1_small.txt
(timeout 300 python bench.py --workload bacteria5 --scale 0.2 --steps 2 --warmup 1 2>&1 | grep -v "amdgpu.ids" | tail -12) > gpurun_out/c2_bench_c5_small.txt
bash tools/rehearsal_diag.sh 0.05 1 p2p 300 "100,1:1;30,1:1" > gpurun_out/c2_rehearsal_small.txt 2>&1
cp gpurun_out/diag_p2p.txt gpurun_out/diag_p2p_small.txt
bash tools/rehearsal_diag.sh 1.0 0 p2p 420 "100,1:1" > gpurun_out/c2_rehearsal_full_p2p.txt 2>&1
