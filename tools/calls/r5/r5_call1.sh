#!/bin/bash
# round 5, call 1: the round-4 kernels with the round-5 bench protocol (per-window median/min, latency probe) and the new GPU test
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
tools/gpu.sh box:r05_a | head -n 8; lap box
tools/gpu.sh bench:r05_a_c3 | cut -c1-600; lap bench3
tools/gpu.sh tests | tail -n 8; lap tests
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_a_c3_bench.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step", "per_step", "box", "value_end_to_end")})
print(j["roofline"]["frac"], j["roofline"].get("pass_a_us_median_min"), j["roofline"].get("pass_b_us_median_min"), j["roofline"].get("frac_at_median"), j["roofline"].get("measured_copy_GBs"))
PY
