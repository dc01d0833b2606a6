#!/bin/bash
# round 5, call 19: what a window costs when every window meets edges the kept set lacks (tools/churn_probe.py)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 400 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 6
