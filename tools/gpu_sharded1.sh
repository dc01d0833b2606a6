#!/bin/bash
# single-GPU checks of the multi-GPU code path: world=1 under torch.distributed.run
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_dist1.log 2>&1
echo "rc=$?" >> gpurun_out/bench_dist1.log
SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench_sharded1.log 2>&1
echo "rc=$?" >> gpurun_out/bench_sharded1.log
tail -n 3 gpurun_out/bench_dist1.log | cut -c1-600; tail -n 6 gpurun_out/bench_sharded1.log | cut -c1-900
