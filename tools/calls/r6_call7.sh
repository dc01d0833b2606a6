#!/bin/bash
# round 6, call 7: why a delta window costs 600 us — paths and per-kernel times of the churn probe's first stream
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
( cd /tmp && rm -rf $GRAFT_REPO_ROOT/$O/prof_churn && CHURN_ONLY_FIRST=1 timeout 500 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_churn -o kt -- python $GRAFT_REPO_ROOT/tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 5
  python $GRAFT_REPO_ROOT/tools/rocpd_stats.py $GRAFT_REPO_ROOT/$O/prof_churn/kt_results.db $GRAFT_REPO_ROOT/$O/r06_c_churn_kernel_stats.txt | head -n 30; rm -rf $GRAFT_REPO_ROOT/$O/prof_churn )
