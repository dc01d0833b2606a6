#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "config2 or edge_cases or overflow_paths or alive or random_small or empty_and_tiny or table_updates or pass_b_second or device_resident" > $O/c4_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 5 $O/c4_pytest.log | grep -v amdgpu
B="--no-cpu-baseline --no-end-to-end --overlap-windows 0 --settle-ms 100"
run() { tag=$1; shift; env "$@" python bench.py $B > $O/c4_$tag.json 2> $O/c4_$tag.err; echo "$tag rc=$?"; }
run t768 SG_K1A=team
run t1024 SG_K1A=team SG_K1A_NT=1024
run tile SG_K1A=tile
run t768b SG_K1A=team
python - <<'PY'
import json
for t in ("t768","t1024","tile","t768b"):
    try:
        j = json.loads(open(f"gpurun_out/c4_{t}.json").read().strip().splitlines()[-1])
        print(t, round(j["ms_per_step"]*1e3,1), "us/window  K1a", round(j["roofline"]["pass_a_us"],1), "K1b", round(j["roofline"]["pass_b_us"],1), "frac", round(j["roofline"]["frac"],4), j["roofline"]["kernels"][0], j["config"]["events_dropped_cap"], j["roofline"]["geometry"]["cache_slots"], j["roofline"]["geometry"]["tile_records"])
    except Exception as e: print(t, "ERR", e, open(f"gpurun_out/c4_{t}.err").read()[-600:])
PY
