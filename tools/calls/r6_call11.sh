#!/bin/bash
# round 6, call 11: development build (knobs compiled out of the shipped library), team kernel for 18-bit endpoints (C5 shard), churn leg again
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "config5 or mfma_dense or k4_gather_launch or row_sort_by_blocks or row_degrees_by or packed_add or new_edges_window" 2>&1 | tail -n 6
python bench.py --config 5 --shard-of 8 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_f_c5_shard.json 2> $O/r06_f_c5_shard.err; echo "c5 shard rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r06_f_bench_c3_gpu_legs.json 2> $O/r06_f_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
def line(p): return json.loads(open(p).read().strip().splitlines()[-1])
j = line("gpurun_out/r06_f_c5_shard.json"); print("c5 shard", round(j["ms_per_step"] * 1e3, 1), j["config"]["events_per_window"], j["config"]["edges_per_window"], round(j["roofline"]["frac"], 4), j["roofline"]["kernels"], [(k["name"], k["us_per_window"]) for k in j["kernels"]])
j = line("gpurun_out/r06_f_bench_c3_gpu_legs.json")
print("C3", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], "frac", round(j["roofline"]["frac"], 4), j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"])
print([(k["name"], k["us_per_window"]) for k in j["kernels"]], "cold", j["warm_windows"]["cold_ms_per_step"])
for c in j["warm_windows"]["churn"] if isinstance(j["warm_windows"]["churn"], list) else [j["warm_windows"]["churn"]]: print(c)
PY
SG_ABLATE=0x100 CHURN_ONLY_FIRST=1 timeout 300 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids | tail -n 4
