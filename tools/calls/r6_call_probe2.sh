#!/bin/bash
# round 6, pass B on one box: v0 = round 5's loop (per-record probe, a round's loads waited for in front of its merges, U = 4),
# v1 = the pair probe loop alone, v2 = pair loop + rolling load buffer (U = 2), v3 = round 5's probe + rolling buffer (U = 2), v4 = v2 at U = 3
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  for v in v0 v1 v2 v3 v4; do
    SG_LIB_DEV=$PWD/alaz_amd/lib/ab_$v.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[$v] /" | cut -c1-200 | tee -a $O/r06_probe_ab2.txt
  done
done
SG_LIB_DEV=$PWD/alaz_amd/lib/ab_v2.so timeout 300 python tools/stamps.py 3 2>&1 | grep -v amdgpu.ids | sed -n '/k1b_stream_merge/,/kw_compact/p' | tee $O/r06_probe_stamps_v2.txt
