#!/bin/bash
# round 6 (late), two steps against the development build of commit 1a61c11 (alaz_amd/lib/ab_head_dev.so), one box, three alternating repetitions:
#  [B up front]   k4_sage_layer fetches the dense tiles' B operands (its 16 columns of W, its share of the projection) in one round trip ahead of
#                 phase 1 instead of four at a time inside the MFMA chain — ab_wpre_dev.so
#  [+ row loads]  phase 1 of the same kernel: a row's five loads issued together, a hub row's block sums eight at a time
# then the K4 / MFMA / window tests on the shipped build, and the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_head_dev.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[head] /" | cut -c1-230 | tee -a $O/r06_wpre2_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_wpre_dev.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[B up front] /" | cut -c1-230 | tee -a $O/r06_wpre2_ab.txt
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[+ row loads] /" | cut -c1-230 | tee -a $O/r06_wpre2_ab.txt
done
tools/gpu.sh "tests:k4,or,mfma,or,hub,or,blocks,or,shards,or,config" | tail -n 12
