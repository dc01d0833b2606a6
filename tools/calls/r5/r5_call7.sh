#!/bin/bash
# round 5, call 7: the K1 path by window size (config 2 on the 16-byte kernels again), warm state only from 2^18 edges; whole suite + both lines
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
tools/gpu.sh tests | tail -n 12; lap tests
tools/gpu.sh bench:r05_g_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-200; lap bench3
tools/gpu.sh bench:r05_g_c2:--config,2,--no-cpu-baseline,--no-end-to-end | cut -c1-200; lap bench2
tools/gpu.sh prof:r05_g:2 | head -n 18; lap prof2
python - <<'PY'
import json
for c in (3, 2):
    j = json.loads(open(f"gpurun_out/r05_g_c{c}_bench.json").read().strip().splitlines()[-1])
    print(c, round(j["ms_per_step"] * 1e3, 1), "median", j["per_step"]["median_ms"], "min", j["per_step"]["min_ms"], "frac", round(j["roofline"]["frac"], 4), [(k["name"], k["us_per_window"]) for k in j["kernels"]], j["warm_windows"]["cold_ms_per_step"], j["box"]["hbm_latency_ns"])
PY
