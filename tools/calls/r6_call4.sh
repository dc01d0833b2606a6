#!/bin/bash
# round 6, call 4: pass A same-box A/B behind the ticket hand-over fix: team (r4) / team1 / team1ov, each in its own process under a timeout
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for v in "" "SG_K1A=team1" "SG_K1A=team1ov" "" "SG_K1A=team1"; do
  SWEEP_STEPS=12 timeout 240 python tools/k1_sweep.py 3 "$v" 2>&1 | grep -v amdgpu.ids | tail -n 2; echo "rc=$?"
done
