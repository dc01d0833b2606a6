#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or edge_cases or overflow or config2 or alive" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 6 gpurun_out/pytest_gpu.log
timeout 600 python tools/k1_sweep.py 3 "" "SG_K1B_U=8" "SG_K1B_THREADS=1024" "SG_K1B_THREADS=256" "SG_K1B_THREADS=256 SG_K1B_U=8" > gpurun_out/sweep_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c3.log
timeout 600 python tools/k1_sweep.py 2 "" "SG_K1B_U=8" "SG_K1B_THREADS=1024" "SG_K1B_THREADS=256" "SG_NP=512" > gpurun_out/sweep_c2.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c2.log
