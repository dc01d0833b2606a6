#!/bin/bash
# round 5, call 23: the sharded entry point at world = 1 under torch.distributed.run on the last tree (call 22 started it without a launcher)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh sharded1:r05_final | tail -n 3
