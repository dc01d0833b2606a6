#!/bin/bash
# round 6: on the sources as they ship (TAG, default r06_ship) — smoke, the default bench line, rocprofv3 kernel stats, the PMC passes on exactly
# these sources (profiles/pmc_k1_c3.json: bench.py's traffic_build_matches), the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
TAG=${1:-r06_ship}
tools/gpu.sh smoke
tools/gpu.sh bench:${TAG}_c3 | cut -c1-300
tools/gpu.sh prof:$TAG:3 | head -n 24
tools/gpu.sh pmc:$TAG:3 | tail -n 6
tools/gpu.sh tests | tail -n 6
