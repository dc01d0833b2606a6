#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for sc in strong weak; do
SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --scaling $sc --verify --settle-ms 100 > $O/c9_sh_$sc.json 2> $O/c9_sh_$sc.err
echo "$sc rc=$?"; tail -n 1 $O/c9_sh_$sc.json | cut -c1-2200; tail -n 3 $O/c9_sh_$sc.err | grep -v amdgpu
done
timeout 900 python bench.py --config 5 --shard-of 8 --no-cpu-baseline --no-end-to-end --settle-ms 100 > $O/c9_c5shard.json 2> $O/c9_c5shard.err; echo "c5shard rc=$?"; tail -n 1 $O/c9_c5shard.json | cut -c1-2500; tail -n 3 $O/c9_c5shard.err | grep -v amdgpu
