#!/bin/bash
# round 6 (late): the chain kernels' mode (C_COLD, C_DELTA_N) read in one round trip instead of two dependent ones — three launches per plain warm
# window are nothing but that latency; against the development build of commit c38d10a (alaz_amd/lib/ab_fin_dev.so), one box, four alternating repetitions
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3 4; do
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_fin_dev.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[two trips] /' | cut -c1-200 | tee -a $O/r06_chainmode_ab.txt
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[one trip] /' | cut -c1-200 | tee -a $O/r06_chainmode_ab.txt
done
