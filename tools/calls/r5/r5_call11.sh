#!/bin/bash
# round 5, call 11: config 5 as one shard of eight, the bench's engine: rows against the oracle + per-kernel durations (the close went 1.5 -> 0.66 ms: why)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
( cd /tmp && rm -rf $GRAFT_REPO_ROOT/$O/prof_c5s && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_c5s -o kt -- python $GRAFT_REPO_ROOT/tools/c5_shard_check.py > $GRAFT_REPO_ROOT/$O/r05_c5_shard_check.log 2>&1 )
grep -v amdgpu.ids $O/r05_c5_shard_check.log | tail -n 8
python tools/rocpd_stats.py $O/prof_c5s/kt_results.db $O/r05_c5_shard_kernel_stats.txt | head -n 24; rm -rf $O/prof_c5s
