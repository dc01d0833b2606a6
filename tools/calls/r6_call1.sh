#!/bin/bash
# round 6, call 1: pass A as two workgroups per CU (k1a_duo_partition) against the team kernel, same box; parity subset under the new kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh box:r06_a | head -n 4
tools/gpu.sh "sweep:3:;SG_K1A=duo;SG_K1A=duo,SG_L2_GLOBAL=1;SG_K1A=duo,SG_K1A_NT=640;SG_K1A=duo,SG_K1A_NT=640,SG_L2_GLOBAL=1;SG_K1A=duo,SG_K1A_NT=512;"
SG_K1A=duo timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3 or warm" 2>&1 | tail -n 8
