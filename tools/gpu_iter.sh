#!/bin/bash
# one iteration on the GPU box: the named tests (K=...), then optional extras
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "${K:-not config5}" 2>&1 | grep -v "^  File\|Extension modules" | tail -15
if [ -n "$SWEEP" ]; then
  timeout 600 python tools/k1_sweep.py 2 "" "SG_SWEEP_HIST=1" 2>&1 | tail -2
  timeout 600 python tools/k1_sweep.py 3 "" "SG_SWEEP_HIST=1" 2>&1 | tail -2
fi
