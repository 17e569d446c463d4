#!/bin/bash
# Rehearsal of bench.py's N>1 path on a 1-GPU box: 2 ranks share GPU 0, gloo gather.
set -e
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 \
    bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo
