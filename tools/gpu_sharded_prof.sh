#!/bin/bash
# rocprof of the forced-sharded path at world = 1 (the multi-GPU code path incl. RCCL calls, on one GPU)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
SG_FORCE_SHARDED=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_sh1" -o sh1 -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 "$GRAFT_REPO_ROOT/bench.py" --gpus 1 --steps 50 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/sh1.log" 2>&1
cd "$GRAFT_REPO_ROOT"
ls gpurun_out/prof_sh1 | head
for f in gpurun_out/prof_sh1/*results.db gpurun_out/prof_sh1/*/*results.db; do [ -f "$f" ] && python tools/rocpd_stats.py "$f" gpurun_out/sh1_kernel_stats.txt | head -40; done
tail -n 2 gpurun_out/sh1.log | cut -c1-300
