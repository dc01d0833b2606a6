#!/bin/bash
# round 6, pass B's workgroups in the order of their partitions' size (largest first, from the window before: kc_prepare's counting sort) against
# block order (SG_K1B_NO_ORDER=1), one box, the development build; then the whole -m gpu suite on the shipped build
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python tools/k1_sweep.py 3 "SG_ABLATE=0" "SG_K1B_NO_ORDER=1" "SG_ABLATE=0" "SG_K1B_NO_ORDER=1" "SG_ABLATE=0" "SG_K1B_NO_ORDER=1" 2>&1 | grep -v amdgpu.ids | cut -c1-220 | tee $O/r06_order_ab.txt
tools/gpu.sh tests
