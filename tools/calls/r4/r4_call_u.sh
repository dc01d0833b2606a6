#!/bin/bash
# round 4: K1 geometry re-swept after pass B lost its degree atomics (is 512 x 2 still the best split?), phase stamps and SQ / MFMA
# counters of the final kernels (the r04_z counters are from the build before k2_deg_hist / the relaxed look-back)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
tools/gpu.sh "sweep:3:;SG_NP=256,SG_HT=4096;SG_NP=1024,SG_SPLIT=1,SG_HT=2048;SG_NP=1024,SG_HT=1024;SG_K1B_U=8;SG_DH_G=64;SG_DH_G=32;SG_K3_SLICES=24;SG_K3_SLICES=48;"; lap sweep
tools/gpu.sh stamps:3 | tail -n 40; lap stamps
tools/gpu.sh sq:r04_zz:3 | tail -n 30; lap sq
tools/gpu.sh mfma:r04_zz:3 | tail -n 8; lap mfma
