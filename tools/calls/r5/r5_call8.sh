#!/bin/bash
# round 5, call 8: plain-mode test, stride-3 K1 stamps, loaded-latency probe; C3 line with everything
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_warm.py -m gpu -x -q > $O/pytest_warm.log 2>&1; rc=$?
grep -v "^  File\|Extension modules\|amdgpu.ids" $O/pytest_warm.log | tail -n 30; lap warm
if [ $rc -ne 0 ]; then exit 0; fi
tools/gpu.sh bench:r05_h_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-200; lap bench3
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_h_c3_bench.json").read().strip().splitlines()[-1])
print(round(j["ms_per_step"] * 1e3, 1), "median", j["per_step"]["median_ms"], "min", j["per_step"]["min_ms"], "frac", round(j["roofline"]["frac"], 4), j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"], j["roofline"]["launches"], [(k["name"], k["us_per_window"]) for k in j["kernels"]], j["warm_windows"]["cold_ms_per_step"], j["box"])
PY
