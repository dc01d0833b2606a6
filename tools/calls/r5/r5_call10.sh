#!/bin/bash
# round 5, call 10: flakiness check — the concurrency tests and the warm-window tests three times over
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
for i in 1 2 3; do
  timeout 900 python -m pytest tests/test_gpu_warm.py tests/test_gpu_parity.py -m gpu -q -k "warm or flusher or threads or in_flight or churn or flush_begin" 2>&1 | tail -n 3
done
