#!/bin/bash
# round 6, pass B's table as buckets of four slots (a probe reads a whole bucket; -DSG_K1B_BUCKET4) against the slot-by-slot table, one box:
# development builds for the sweep, then the warm + parity tests on a shipped-configuration build of the bucket form
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[slots] /" | cut -c1-200 | tee -a $O/r06_bucket_ab.txt
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_b4.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/^/[buckets of 4] /" | cut -c1-200 | tee -a $O/r06_bucket_ab.txt
done
SG_LIB=$PWD/alaz_amd/lib/ship_b4.so timeout 1500 python -m pytest tests/test_gpu_warm.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest_bucket.log 2>&1; tail -n 3 $O/pytest_bucket.log
