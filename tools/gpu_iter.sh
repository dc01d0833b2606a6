#!/bin/bash
# one iteration on the GPU box: the named tests (K=...), then optional sweeps (SWEEP="variant strings" separated by ;)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -x -q -k "${K:-not config5}" 2>&1 | grep -v "^  File\|Extension modules" | tail -15
if [ -n "$SWEEP" ]; then
  IFS=';' read -ra V <<< "$SWEEP"
  timeout 600 python tools/k1_sweep.py 3 "${V[@]}" 2>&1 | grep -v amdgpu.ids
  timeout 600 python tools/k1_sweep.py 2 "${V[@]}" 2>&1 | grep -v amdgpu.ids
fi
