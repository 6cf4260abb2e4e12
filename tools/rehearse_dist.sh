#!/bin/bash
# Single-GPU rehearsal of the multi-GPU launch path: torch.distributed.run with one rank, torch's nccl process group,
# unique-id broadcast, RCCL communicators + split, grouped self send/recv on the side streams, all-reduce in GMRES.
set -e
cd "$(dirname "$0")/.."
export LSFC_BENCH_N=${1:-256} LSFC_DIST_FORCE_COMM=1 LSFC_DIST_FORCE_OVERLAP=1 LSFC_DIST_CHUNKS=4
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --no-cpu-baseline
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 tools/rehearse_dist_gmres.py
