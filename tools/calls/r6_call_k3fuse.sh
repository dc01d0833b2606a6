#!/bin/bash
# round 6: the in-statistics' reduction inside k3_node_features (one launch less per window) against the two-launch form (SG_K3_NO_FUSE=1), one box,
# the development build; then the whole -m gpu suite on the shipped build
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python tools/k1_sweep.py 3 "SG_ABLATE=0" "SG_K3_NO_FUSE=1" "SG_ABLATE=0" "SG_K3_NO_FUSE=1" "SG_ABLATE=0" "SG_K3_NO_FUSE=1" 2>&1 | grep -v amdgpu.ids | cut -c1-230 | tee $O/r06_k3fuse_ab.txt
tools/gpu.sh tests
