#!/bin/bash
# round 5, call 3: warm windows — fused capture, two workgroups per CU in kw_compact, wave-uniform statistics; one stream vs two
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests/test_gpu_warm.py -m gpu -x -q > $O/pytest_warm.log 2>&1; rc=$?
grep -v "^  File\|Extension modules\|amdgpu.ids" $O/pytest_warm.log | tail -n 30; lap warm
if [ $rc -ne 0 ]; then exit 0; fi
for aux in 1 0; do
  export SG_WARM_AUX=$aux
  tools/gpu.sh bench:r05_c_aux${aux}_c3:--no-cpu-baseline,--no-end-to-end | cut -c1-200
  tools/gpu.sh prof:r05_c_aux${aux}:3 | head -n 26; lap aux$aux
done
unset SG_WARM_AUX
python - <<'PY'
import json
for a in (1, 0):
    j = json.loads(open(f"gpurun_out/r05_c_aux{a}_c3_bench.json").read().strip().splitlines()[-1])
    print(a, j["ms_per_step"], j["per_step"]["median_ms"], j["per_step"]["min_ms"], j["warm_windows"]["cold_ms_per_step"], j["roofline"]["frac"], [(k["name"], k["us_per_window"]) for k in j["kernels"]])
PY
tools/gpu.sh tests | tail -n 6; lap tests
