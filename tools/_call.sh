cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29547 bench.py --gpus 2 --workload bacteria5 --same-device --backend gloo --steps 2 --warmup 1 --verify --no-cpu-baseline 2>&1 | grep -v "Warning\|amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*\|^$" | cut -c1-1200 | tail -12) > gpurun_out/c59_c5_2rank.txt
