#!/bin/bash
# round 6: behind the last kernel edit (a dropped image key no longer keeps its partition cold) — the whole -m gpu suite, smoke,
# the PMC passes on exactly these sources (pmc_k1_c3.json), the default bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh smoke
tools/gpu.sh bench:r06_last_c3 | cut -c1-300
tools/gpu.sh pmc:r06_last:3 | tail -n 6
tools/gpu.sh tests | tail -n 6
