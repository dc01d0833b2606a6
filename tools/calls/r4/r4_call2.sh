#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "config2 or edge_cases or overflow_paths or alive or random_small or empty_and_tiny or table_updates or pass_b_second or device_resident" > $O/c2_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 15 $O/c2_pytest.log | grep -v amdgpu
B="--no-cpu-baseline --no-end-to-end --overlap-windows 0 --settle-ms 100"
for k in team tile; do SG_K1A=$k python bench.py $B > $O/c2_$k.json 2> $O/c2_$k.err; echo "$k rc=$?"; done
python - <<'PY'
import json
for t in ("team","tile"):
    try:
        j = json.loads(open(f"gpurun_out/c2_{t}.json").read().strip().splitlines()[-1])
        print(t, round(j["ms_per_step"]*1e3,1), "us/window  K1a", round(j["roofline"]["pass_a_us"],1), "K1b", round(j["roofline"]["pass_b_us"],1), "frac", round(j["roofline"]["frac"],4), j["roofline"]["kernels"], j["config"]["events_dropped_cap"], j["roofline"]["geometry"]["cache_slots"])
    except Exception as e: print(t, "ERR", e, open(f"gpurun_out/c2_{t}.err").read()[-600:])
PY
SG_K1A=team timeout 300 python tools/stamps.py 3 2>&1 | grep -v amdgpu | tail -n 22
