#!/bin/bash
# round 6 (late): the in-statistics' slice count again, now that k3_node_features sums the slices (more slices = fewer trips in k3_in_part's scan
# and every CU busy, but more partials for the features to read); one box, the development build, two repetitions
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python tools/k1_sweep.py 3 "SG_ABLATE=0" "SG_K3_SLICES=24" "SG_K3_SLICES=40" "SG_K3_SLICES=48" "SG_K3_SLICES=51" "SG_ABLATE=0" "SG_K3_SLICES=24" "SG_K3_SLICES=40" "SG_K3_SLICES=48" "SG_K3_SLICES=51" 2>&1 | grep -v amdgpu.ids | cut -c1-230 | tee $O/r06_k3slices_ab.txt
