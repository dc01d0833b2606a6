#!/bin/bash
# round 6, call 8: phase stamps of kw_compact on a delta window
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
SG_ABLATE=0x100 CHURN_ONLY_FIRST=1 timeout 500 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 8
