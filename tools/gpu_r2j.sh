#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do echo "== run $i"; timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config1 or edge_cases or empty or table_updates" 2>&1 | grep -v "^  File\|Extension modules" | tail -1; done
timeout 900 python -m pytest tests -m gpu -x -q -k "not config5" 2>&1 | grep -v "^  File\|Extension modules" | tail -2
timeout 600 python tools/k1_sweep.py 3 "" "SG_NO_HOT=1" "SG_CT=1024" "SG_CT=512" > gpurun_out/sweep_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c3.log
timeout 600 python tools/k1_sweep.py 2 "" "SG_NO_HOT=1" "SG_CT=1024" > gpurun_out/sweep_c2.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c2.log
