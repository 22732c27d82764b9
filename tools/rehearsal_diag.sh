#!/bin/bash
# Two ranks sharing ONE device at the metric's own size, with stage markers from every rank and Python stacks after a stall:
# tools/rehearsal_diag.sh [scale] [sampling] [p2p|collective] [limit seconds] [sub-records]
# (gloo as control plane; --verify compares every gathered vector with the single-rank one; the JSON line is the N=2 record)
SC=${1:-1.0}; SA=${2:-0}; CM=${3:-p2p}; LIM=${4:-420}; SUB=${5-100,1:1}
export MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
timeout $LIM python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --E 0 --workload grch38 --scale $SC --sampling $SA \
  --same-device --backend gloo --comm $CM --watchdog 250 --steps 3 --warmup 1 --verify --trace 150 --sub "$SUB" --no-cpu-baseline > gpurun_out/diag_${CM}.txt 2>&1
echo "rc=$?" >> gpurun_out/diag_${CM}.txt
grep -v "Warning\|amdgpu.ids\|socket.cpp\|OMP_NUM\|\*\*\*\*\|^$" gpurun_out/diag_${CM}.txt | cut -c1-2000 | tail -70
