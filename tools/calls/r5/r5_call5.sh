#!/bin/bash
# round 5, call 5: where kw_compact's time goes (phase stamps), and the warm path against the rebuild-every-window engine on the SAME box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_warm.py -m gpu -x -q > $O/pytest_warm.log 2>&1; rc=$?
grep -v "^  File\|Extension modules\|amdgpu.ids" $O/pytest_warm.log | tail -n 30; lap warm
if [ $rc -ne 0 ]; then exit 0; fi
tools/gpu.sh stamps:3 | grep -A12 "kw_compact\|geometry" | head -n 40; lap stamps
for w in 1; do
  export SG_WARM=$w
  tools/gpu.sh bench:r05_e_warm${w}_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-200
  tools/gpu.sh prof:r05_e_warm${w}:3 | head -n 24; lap warm$w
done
unset SG_WARM
python - <<'PY'
import json
for a in (1,):
    j = json.loads(open(f"gpurun_out/r05_e_warm{a}_c3_bench.json").read().strip().splitlines()[-1])
    print(a, j["ms_per_step"], j["per_step"]["median_ms"], j["per_step"]["min_ms"], j["roofline"]["frac"], [(k["name"], k["us_per_window"]) for k in j["kernels"]], j["box"])
PY
