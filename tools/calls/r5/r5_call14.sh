#!/bin/bash
# round 5, call 14: mid-size graphs (one shard of 8 / 4 of C3): the fused K4 layer against gather + dense launches
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for w in 8 4; do for f in 0 1; do
  SG_K4_FUSED=$f python bench.py --config 3 --shard-of $w --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r05_k4_${w}_$f.json 2>/dev/null
  python - $w $f <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r05_k4_{sys.argv[1]}_{sys.argv[2]}.json").read().strip().splitlines()[-1])
print("shard-of", sys.argv[1], "fused" if sys.argv[2] == "1" else "split", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], [(k["name"], k["us_per_window"]) for k in j["kernels"] if k["name"] in ("K4", "K5", "K2")])
PY
done; done
