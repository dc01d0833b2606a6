#!/bin/bash
# round 6, call 16: delta windows without the returning degree atomic in pass B (the delta scatter ranks by cursor): warm tests + the churn leg per kernel group
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 1200 python -m pytest tests/test_gpu_warm.py -m gpu -q 2>&1 | tail -n 4
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_i_bench_c3_gpu_legs.json 2> $O/r06_i_bench.err; echo "bench rc=$?"; tail -n 3 $O/r06_i_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_i_bench_c3_gpu_legs.json").read().strip().splitlines()[-1])
print("C3", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"])
for c in j["warm_windows"]["churn"] if isinstance(j["warm_windows"]["churn"], list) else [j["warm_windows"]["churn"]]: print(c if "error" in c else (c["new_edges_per_window"], c["ms_per_window_median"], c["ms_same_windows_edges_known"], c["vs_same_windows_edges_known"], c["us_per_kernel_group"]))
PY
