#!/bin/bash
# round 6 (late): k4_gather's workgroup size — 4 rows (waves) per workgroup as shipped against 2 and 1 (development builds with -DK4G_ROWS=2 / 1:
# alaz_amd/lib/ab_rows2.so, ab_rows1.so), one box, two alternating repetitions
# (K4G_ROWS was made overridable for this call only; no difference, the #ifndef is not in the tree)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2; do
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[4 rows] /' | cut -c1-200 | tee -a $O/r06_rows_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_rows2.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[2 rows] /' | cut -c1-200 | tee -a $O/r06_rows_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_rows1.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[1 row] /' | cut -c1-200 | tee -a $O/r06_rows_ab.txt
done
