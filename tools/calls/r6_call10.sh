#!/bin/bash
# round 6, call 10: the bench line with the churn leg (no CPU baseline / end-to-end: GPU legs only), the warm tests again behind the trimmed delta path
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_e_bench_c3_gpu_legs.json 2> $O/r06_e_bench.err; echo "bench rc=$?"; tail -n 3 $O/r06_e_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_e_bench_c3_gpu_legs.json").read().strip().splitlines()[-1])
print("C3", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], "frac", round(j["roofline"]["frac"], 4), j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"])
print([(k["name"], k["us_per_window"]) for k in j["kernels"]], "cold", j["warm_windows"]["cold_ms_per_step"])
for c in j["warm_windows"]["churn"] if isinstance(j["warm_windows"]["churn"], list) else [j["warm_windows"]["churn"]]: print(c)
PY
timeout 900 python -m pytest tests/test_gpu_warm.py -m gpu -q -x 2>&1 | tail -n 3
