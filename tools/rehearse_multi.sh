#!/bin/bash
# Rehearsal of bench.py's N>1 path on a 1-GPU box: N ranks (default 2, at most 4) share GPU 0, gloo gather.
set -e
N=${1:-2}
python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus $N --steps 3 --warmup 1 --backend gloo
