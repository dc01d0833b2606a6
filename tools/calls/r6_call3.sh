#!/bin/bash
# round 6, call 3: team1 alone under a short timeout (call 2's sweep hung behind the default variant)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
SWEEP_STEPS=4 timeout 240 python tools/k1_sweep.py 3 "SG_K1A=team1" 2>&1 | grep -v amdgpu.ids | tail -n 5; echo "rc=$?"
SWEEP_STEPS=4 timeout 120 python tools/k1_sweep.py 3 "SG_K1A=team1ov" "" 2>&1 | grep -v amdgpu.ids | tail -n 5; echo "rc=$?"
