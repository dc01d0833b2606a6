#!/bin/bash
# round 6, call 2: pass A variants on one box: team (r4) / team1 (16 waves, one group per tile) / team1ov / duo
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh "sweep:3:;SG_K1A=team1;SG_K1A=team1ov;SG_K1A=duo;;SG_K1A=team1"
SG_K1A=team1 timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3 or warm" 2>&1 | tail -n 4
