#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config3 or edge_cases or overflow" > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 12 gpurun_out/pytest_gpu.log
timeout 1200 python tools/k1_sweep.py 3 "" "SG_ABLATE=0x1" "SG_ABLATE=0x8" "SG_ABLATE=0x2" "SG_ABLATE=0x4" "SG_ABLATE=0x10" \
  "SG_NP=1024 SG_HT=2048" "SG_NP=1024 SG_HT=2048 SG_ABLATE=0x1" "SG_NP=4096" "SG_NP=4096 SG_HT=512" "SG_NP=4096 SG_HT=512 SG_K1B_U=8" \
  "SG_K1B_U=8" "SG_K1B_THREADS=1024" "SG_K1B_THREADS=1024 SG_K1B_U=8" "SG_K1B_THREADS=256 SG_K1B_U=8" "SG_CT=1024" "SG_CT=256" > gpurun_out/sweep_c3.log 2>&1
cat gpurun_out/sweep_c3.log | grep -v amdgpu.ids
timeout 600 python tools/k1_sweep.py 2 "" "SG_ABLATE=0x1" "SG_ABLATE=0x8" "SG_ABLATE=0x2" "SG_NP=512" "SG_NP=1024" "SG_K1B_U=8" "SG_K1B_THREADS=1024" "SG_NP=512 SG_K1B_THREADS=256 SG_K1B_U=8" > gpurun_out/sweep_c2.log 2>&1
cat gpurun_out/sweep_c2.log | grep -v amdgpu.ids
