#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[all lanes read] /" | cut -c1-200 | tee -a $O/r06_maskedreads_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_mr.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[masked reads] /" | cut -c1-200 | tee -a $O/r06_maskedreads_ab.txt
done
