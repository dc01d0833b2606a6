#!/bin/bash
# round 6, call 6: the whole -m gpu suite behind the delta windows and the new pass A; the churn probe
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 600 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_b_churn_probe.txt
tools/gpu.sh tests | tail -n 30
