#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[oldest first] /" | cut -c1-200 | tee -a $O/r06_prio_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_prio.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[most work left first] /" | cut -c1-200 | tee -a $O/r06_prio_ab.txt
done
SG_LIB_DEV=$PWD/alaz_amd/lib/ab_prio.so timeout 300 python tools/stamps.py 3 2>&1 | grep -v amdgpu.ids | sed -n '/k1b_stream_merge/,/kw_compact/p' | cut -c1-400 | tee $O/r06_prio_stamps.txt
