#!/bin/bash
# round 6, call 9: delta windows as merge-path chunks — the warm-window tests, then the churn probe with kw_compact's phase stamps
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests/test_gpu_warm.py -m gpu -q -x 2>&1 | tail -n 6
SG_ABLATE=0x100 CHURN_ONLY_FIRST=1 timeout 500 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 5
timeout 600 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tee $O/r06_d_churn_probe.txt
