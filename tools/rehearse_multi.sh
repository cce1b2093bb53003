#!/bin/bash
# Rehearse bench.py's multi-rank paths on a ONE-GPU box (the driver runs the real
# 2/4/8-GPU scaling bench).  (1) one-rank RCCL group: replicated, sharded (+gather)
# layouts through the nccl backend; (2) two ranks on cuda:0 over gloo: the replicated
# layout's partitioning, barrier and max-over-ranks timing.
set -o pipefail
L=${1:-128}
export QS_BENCH_FORCE_DIST=1
for args in "--layout replicated" "--layout replicated --gather" "--layout sharded" "--layout sharded --gather"; do
  echo "== 1 rank, RCCL, $args"
  timeout -k 10 300 python bench.py --gpus 1 --orbitals $L --steps 3 --warmup 1 --no-cpu-baseline --no-probes $args 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['config']['layout'], d['parity'])" || exit 1
done
unset QS_BENCH_FORCE_DIST
echo "== 2 ranks on cuda:0 over gloo, replicated"
QS_BENCH_SINGLE_DEVICE=1 QS_BENCH_BACKEND=gloo timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --orbitals $L --steps 3 --warmup 1 --no-cpu-baseline --no-probes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['layout'], d['parity'])"
