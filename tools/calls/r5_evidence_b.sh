#!/bin/bash
# round 5, evidence refresh after the last kernel change (warm pass B says "cold" at once): suite, C3 line (GPU legs), kernel stats, PMC, SQ counters
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
TAG=${1:-r05_zz}
timeout 600 python -m pytest tests/test_gpu_warm.py -m gpu -x -q 2>&1 | tail -n 3; lap warm
tools/gpu.sh bench:${TAG}_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-200; lap bench3
tools/gpu.sh prof:$TAG:3 | head -n 22; lap prof3
tools/gpu.sh pmc:$TAG:3 | tail -n 12; lap pmc3
tools/gpu.sh sq:$TAG:3 > /dev/null; lap sq3
tools/gpu.sh tests | tail -n 5; lap tests
python - $TAG <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/{sys.argv[1]}_c3_bench.json").read().strip().splitlines()[-1])
print(round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], j["per_step"]["min_ms"], round(j["roofline"]["frac"], 4), j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"], j["roofline"]["traffic_build_matches"], [(k["name"], k["us_per_window"]) for k in j["kernels"]], j["warm_windows"]["cold_ms_per_step"], j["box"].get("device"))
PY
