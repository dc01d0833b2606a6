#!/bin/bash
# round 5, call 12: what the driver runs at round end — smoke(), then its bench command
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 3
T0=$(date +%s)
python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r05_driver_like.json 2> $O/r05_driver_like.err; echo "bench rc=$? in $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_driver_like.json").read().strip().splitlines()[-1])
print({k: j[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")})
print(j["config"]["workload"]); print(j["roofline"]["frac"], j["roofline"]["traffic_build_matches"], j["roofline"]["launches"], j["per_step"]["median_ms"], j["cpu_baseline"]["value"], j["cpu_baseline"]["cores"], j["cpu_baseline"].get("sample"), j["value_end_to_end"])
PY
