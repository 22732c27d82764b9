#!/bin/bash
# Two ranks sharing ONE device (gloo control plane) against one rank, same workload: what the multi-GPU data path costs beyond the
# search itself (interleaved chunks, strided clear / finalize, IPC peer copies).  tools/rehearsal_2rank.sh [scale] [sampling] [p2p|collective]
SC=${1:-0.25}; SA=${2:-1}; CM=${3:-p2p}; KK=${4:-30}; EE=${5:-0}   # [K E]: e.g. 100 1 = config C4
export MASTER_ADDR=127.0.0.1
echo "== N=1 scale $SC sampling $SA"; timeout 900 python bench.py --K $KK --E $EE --workload grch38 --scale $SC --sampling $SA --no-cpu-baseline --no-counters --no-host-rate --sub "" --steps 10 --warmup 2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('k-mers/s %.4g  ms/step %.2f  kernel ms %.2f' % (d['value'], d['ms_per_step'], d['roofline']['kernel_ms']))"
echo "== N=2 same device, --comm $CM"; timeout 1200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --K $KK --E $EE --workload grch38 --scale $SC --sampling $SA --same-device --backend gloo --comm $CM --watchdog 300 --steps 10 --warmup 2 --verify --sub "" --no-cpu-baseline --no-counters 2>&1 | grep -v Warning | grep "verify\|^{" | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('k-mers/s %.4g  ms/step %.2f  per-rank search ms %s  per-rank wait ms %s  imbalance %.3f  %s' % (d['value'], d['ms_per_step'], [round(x, 2) for x in d['per_rank_search_ms']], [round(x, 2) for x in d['per_rank_comm_wait_ms']], d['shard_imbalance'], d['comm']))
    else: print(l.strip())"
