#!/bin/bash
# round 6, pass B's probe loop: the pair loop (this build) against round 5's per-record loop (ab_dev_r5probe.so: the same sources
# with -DSG_K1B_PROBE_R5), and one table per partition (SG_SPLIT=1) on the warm path, on one box; then the warm + parity tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
OLD=$PWD/alaz_amd/lib/ab_dev_r5probe.so
for rep in 1 2; do
  SG_LIB_DEV=$OLD timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" "SG_SPLIT=1 SG_K1B_U=4" 2>&1 | grep -v amdgpu.ids | sed 's/^/[r5 probe] /' | tee -a $O/r06_probe_ab.txt
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" "SG_SPLIT=1 SG_K1B_U=4" 2>&1 | grep -v amdgpu.ids | sed 's/^/[pair loop] /' | tee -a $O/r06_probe_ab.txt
done
timeout 1500 python -m pytest tests/test_gpu_warm.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -n 5
SG_SPLIT=1 SG_K1B_U=4 timeout 300 python tools/stamps.py 3 2>&1 | grep -v amdgpu.ids | sed -n '/k1b_stream_merge/,/kw_compact/p' | tee $O/r06_probe_stamps_split1.txt
