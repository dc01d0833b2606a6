#!/bin/bash
# round 6 (late), two steps against the development build of commit 1a61c11 (alaz_amd/lib/ab_head_dev.so), one box, three alternating repetitions:
#  [eight lanes]  k3_node_features with eight lanes per node summing k3_in_part's slices (one round trip, four times the workgroups) — ab_k3lanes_dev.so
#  [+ tile loads] k4_sage_layer<PRE>'s phase 1 as one round trip for the whole tile (every thread an element; the hub rows' block sums eight at a time)
# then the whole -m gpu suite on the shipped build
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_head_dev.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[pair of lanes] /" | cut -c1-230 | tee -a $O/r06_k3lanes_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_k3lanes_dev.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[eight lanes] /" | cut -c1-230 | tee -a $O/r06_k3lanes_ab.txt
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[+ tile loads] /" | cut -c1-230 | tee -a $O/r06_k3lanes_ab.txt
done
tools/gpu.sh tests | tail -n 12
