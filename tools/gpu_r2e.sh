#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
SG_ABLATE=0x100 timeout 600 python tools/stamps.py 3 > gpurun_out/stamps_c3.log 2>&1; grep -v amdgpu gpurun_out/stamps_c3.log
SG_ABLATE=0x110 timeout 600 python tools/stamps.py 3 > gpurun_out/stamps_c3_noadd.log 2>&1; grep -v amdgpu gpurun_out/stamps_c3_noadd.log | tail -7
SG_K1B_U=8 SG_ABLATE=0x100 timeout 600 python tools/stamps.py 3 > gpurun_out/stamps_c3_u8.log 2>&1; grep -v amdgpu gpurun_out/stamps_c3_u8.log | tail -7
timeout 600 python tools/k1_sweep.py 3 "" "SG_K1B_U=8" "SG_K1B_THREADS=512" > gpurun_out/sweep_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c3.log
