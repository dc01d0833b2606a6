#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 12 gpurun_out/pytest_gpu.log
timeout 900 python tools/k1_sweep.py 3 "" "SG_ABLATE=0x1" "SG_ABLATE=0x8" "SG_ABLATE=0x10" "SG_CT=512" "SG_K1B_U=8" "SG_K1B_THREADS=512" "SG_K1B_THREADS=512 SG_K1B_U=8" "SG_NP=2048 SG_HT=1024" > gpurun_out/sweep_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c3.log
timeout 600 python tools/k1_sweep.py 2 "" "SG_ABLATE=0x1" "SG_ABLATE=0x8" "SG_NP=512" "SG_K1B_THREADS=1024" "SG_K1B_THREADS=1024 SG_K1B_U=8" > gpurun_out/sweep_c2.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c2.log
