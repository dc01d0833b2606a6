#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "histogram" 2>&1 | grep -v "^  File\|Extension modules" | tail -25
timeout 900 python -m pytest tests -m gpu -x -q -k "not config5 and not histogram" 2>&1 | grep -v "^  File\|Extension modules" | tail -3
