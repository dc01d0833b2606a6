#!/bin/bash
# round 6, call 13: pass B pair probe vs one record at a time, two shipped-config builds on one box; the in-flight test's failure
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for v in "" "SG_LIB=$GRAFT_REPO_ROOT/alaz_amd/lib/libservicegraph_old.so" "" "SG_LIB=$GRAFT_REPO_ROOT/alaz_amd/lib/libservicegraph_old.so"; do
  SWEEP_STEPS=12 timeout 240 python tools/k1_sweep.py 3 "$v" 2>&1 | grep -v amdgpu.ids | tail -n 1 | cut -c1-330
done
timeout 900 python -m pytest tests/test_gpu_warm.py -m gpu -q -x -k "three_windows or eight_logical" 2>&1 | tail -n 30
