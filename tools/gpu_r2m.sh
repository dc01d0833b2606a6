#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q -k "not config5" 2>&1 | grep -v "^  File\|Extension modules" | tail -4
timeout 300 python tools/k1_sweep.py 3 "" 2>&1 | tail -1
timeout 300 python tools/k1_sweep.py 2 "" 2>&1 | tail -1
