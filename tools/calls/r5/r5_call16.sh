#!/bin/bash
# round 5, call 16: the compaction ordered by ticket (2 M kept edges: 977 chunks > 768 resident) — its own test, bounded
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_warm.py -m gpu -x -q -k "ticket" 2>&1 | grep -v "amdgpu.ids" | tail -n 15
