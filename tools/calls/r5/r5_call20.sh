#!/bin/bash
# round 5, call 20: the device's note to the host (which path the windows took) and the policy on it: warm tests, then the churn probe again
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_warm.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 12
timeout 400 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 5
