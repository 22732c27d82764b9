#!/bin/bash
# N>1 path of bench.py rehearsed on ONE GPU: two ranks, both on cuda:0, gloo (RCCL refuses two ranks per device).
export TMPDIR=/tmp
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --same-device --scale 0.2 2>&1 | tail -4 | cut -c1-900
echo "== verify gathered result equals single-rank result"
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 tools/multi_check.py 2>&1 | tail -3
