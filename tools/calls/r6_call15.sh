#!/bin/bash
# round 6, call 15: where a delta window's extra time goes (per kernel group), the fall-back test
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_h_bench_c3_gpu_legs.json 2> $O/r06_h_bench.err; echo "bench rc=$?"; tail -n 3 $O/r06_h_bench.err
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06_h_bench_c3_gpu_legs.json").read().strip().splitlines()[-1])
for c in j["warm_windows"]["churn"] if isinstance(j["warm_windows"]["churn"], list) else [j["warm_windows"]["churn"]]: print(c if "error" in c else (c["new_edges_per_window"], c["ms_per_window_median"], c["ms_same_windows_edges_known"], c["us_per_kernel_group"]))
PY
timeout 900 python -m pytest tests/test_gpu_warm.py -m gpu -q -x -k "key_budget or pod_gets" 2>&1 | tail -n 5
