#!/bin/bash
# round 5, call 2: warm windows — their own tests first; if green the whole suite, the C3 line and kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_warm.py -m gpu -x -q > $O/pytest_warm.log 2>&1; rc=$?
grep -v "^  File\|Extension modules\|amdgpu.ids" $O/pytest_warm.log | tail -n 40; lap warm
if [ $rc -ne 0 ]; then exit 0; fi
tools/gpu.sh bench:r05_b_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-300; lap bench3
tools/gpu.sh prof:r05_b:3 | head -n 26; lap prof3
tools/gpu.sh tests | tail -n 8; lap tests
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_b_c3_bench.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step", "per_step", "warm_windows")})
print(j["roofline"]["frac"], j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"], [ (k["name"], k["us_per_window"]) for k in j["kernels"]])
PY
