#!/bin/bash
# round 6, call 5: delta windows — the warm-window tests (new expectations), then the C3 / C2 parity tests and a sweep line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests/test_gpu_warm.py -m gpu -q -x 2>&1 | tail -n 25
timeout 600 python -m pytest tests -m gpu -q -x -k "config2 or config3" 2>&1 | tail -n 5
SWEEP_STEPS=12 timeout 240 python tools/k1_sweep.py 3 "" 2>&1 | grep -v amdgpu.ids | tail -n 2
