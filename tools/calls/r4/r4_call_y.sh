#!/bin/bash
# round 4, last call: the packed pass B as default — kernel stats + PMC + bench line (GPU legs only) first, then the full -m gpu suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
tools/gpu.sh bench:r04_zzz_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-300; lap bench3
tools/gpu.sh prof:r04_zzz:3 | head -n 22; lap prof3
tools/gpu.sh pmc:r04_zzz:3 | tail -n 14; lap pmc3
tools/gpu.sh tests | tail -n 5; lap tests
