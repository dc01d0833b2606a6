#!/bin/bash
# round 6, call 19: delta windows with the rank atomic back in pass B and the new-edge FLAG as a plain store: warm tests, churn leg, churn probe
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 1200 python -m pytest tests/test_gpu_warm.py -m gpu -q 2>&1 | tail -n 4
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_j_bench_c3_gpu_legs.json 2> $O/r06_j_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_j_bench_c3_gpu_legs.json").read().strip().splitlines()[-1])
print("C3", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"])
for c in j["warm_windows"]["churn"] if isinstance(j["warm_windows"]["churn"], list) else [j["warm_windows"]["churn"]]: print(c if "error" in c else (c["new_edges_per_window"], c["ms_per_window_median"], c["ms_same_windows_edges_known"], c["vs_same_windows_edges_known"], c["us_per_kernel_group"]))
PY
CHURN_ONLY_FIRST=1 timeout 300 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 3
