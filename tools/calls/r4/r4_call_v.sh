#!/bin/bash
# round 4: does k2_deg_hist pay on small graphs?  C2 (66 k-edge capacity) with and without it; the kc_prepare / K2 kernels of C2 by rocprofv3
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh "sweep:2:;SG_DH_G=0;;SG_DH_G=0"
