#!/bin/bash
# round 6, call 17: the randomised delta / growing-cluster test, then the whole -m gpu suite on the last kernels
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests/test_gpu_warm.py -m gpu -q -x -k "random_windows" 2>&1 | tail -n 15
tools/gpu.sh tests | tail -n 8
