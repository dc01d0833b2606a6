#!/bin/bash
# round 6, call 14: the kept state in compact node ids (a new pod / Host label no longer costs a rebuild) — warm tests, config tests, sharded, churn, a bench line
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 1200 python -m pytest tests/test_gpu_warm.py -m gpu -q 2>&1 | tail -n 25
timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3 or config4 or config5_one or join_table or random or alive" 2>&1 | tail -n 5
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_g_bench_c3_gpu_legs.json 2> $O/r06_g_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_g_bench_c3_gpu_legs.json").read().strip().splitlines()[-1])
print("C3", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], "frac", round(j["roofline"]["frac"], 4), j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"])
print([(k["name"], k["us_per_window"]) for k in j["kernels"]], "cold", j["warm_windows"]["cold_ms_per_step"])
for c in j["warm_windows"]["churn"]: print(c["new_edges_per_window"], c["ms_per_window_median"], c["ms_same_windows_edges_known"], c["vs_same_windows_edges_known"], c["paths"])
PY
