#!/bin/bash
# round 5, call 22: the last tree the way the driver runs it — smoke(), the default bench line, the sharded entry point at world = 1
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2
timeout 200 python bench.py > gpurun_out/r05_final_bench_c3.json 2> gpurun_out/r05_final_bench_c3.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05_final_bench_c3.json"))
print({k: j[k] for k in ("value", "ms_per_step", "scaling")}, j["roofline"]["frac"], j["roofline"].get("traffic_build_matches"), j.get("per_step"))
PY
SG_FORCE_SHARDED=1 SG_BENCH_ONE_MODE=1 timeout 150 python bench.py --steps 20 --warmup 5 > gpurun_out/r05_final_sharded1.json 2> gpurun_out/r05_final_sharded1.err; echo "sharded rc=$?"
head -c 400 gpurun_out/r05_final_sharded1.json
