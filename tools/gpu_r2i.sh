#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not config5" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 6 gpurun_out/pytest_gpu.log
timeout 600 python tools/k1_sweep.py 3 "" "SG_NO_HOT=1" "SG_CT=1024" "SG_CT=512" > gpurun_out/sweep_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c3.log
timeout 600 python tools/k1_sweep.py 2 "" "SG_NO_HOT=1" "SG_CT=1024" > gpurun_out/sweep_c2.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c2.log
