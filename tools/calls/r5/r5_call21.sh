#!/bin/bash
# round 5, call 21: the FETCH / WRITE passes once more (comment-only change in servicegraph.hip since r05_zz: the hash bench.py compares covers it)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh pmc:r05_zz:3 | tail -n 8
