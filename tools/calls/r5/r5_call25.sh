#!/bin/bash
# round 5, call 25: the end-to-end feed from pageable memory with 16 and 12 feeder threads (8 is the default) on one box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp SG_BENCH_E2E=pageable
mkdir -p gpurun_out
for f in 16 8 12; do
  timeout 100 python bench.py --feeders $f --no-cpu-baseline --overlap-windows 0 --steps 10 --warmup 3 --settle-ms 50 > gpurun_out/r05_feed$f.json 2> gpurun_out/r05_feed$f.err; echo "rc=$?"
  python - $f <<'PY'
import json, sys
j = json.load(open(f"gpurun_out/r05_feed{sys.argv[1]}.json")); e = j["end_to_end"]
print("feeders", sys.argv[1], e["events_per_s"], e["ms_per_window"], e["frac_of_pcie_bound"], e["ring_full_retries"], e["pcie_measured_GBs"])
PY
done
