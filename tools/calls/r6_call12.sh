#!/bin/bash
# round 6, call 12: pass B with the two records of a load probed together, A/B on one box (development build: SG_ABLATE 0x40000 = the old form)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for v in "SG_ABLATE=0" "SG_ABLATE=0x40000" "SG_ABLATE=0" "SG_ABLATE=0x40000"; do
  SWEEP_STEPS=12 timeout 240 python tools/k1_sweep.py 3 "$v" 2>&1 | grep -v amdgpu.ids | tail -n 1
done
timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3 or warm or packed_add" 2>&1 | tail -n 3
