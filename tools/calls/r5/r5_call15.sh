#!/bin/bash
# round 5, call 15: does the power-management level explain the two kinds of box?  the C3 GPU legs at 'auto', then at 'high'
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
one() { python bench.py --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/r05_pl_$1.json 2>/dev/null; python - $1 <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r05_pl_{sys.argv[1]}.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], "K1a", round(j["roofline"]["pass_a_us"], 1), "K1b", round(j["roofline"]["pass_b_us"], 1), "frac", round(j["roofline"]["frac"], 4), [(k["name"], k["us_per_window"]) for k in j["kernels"] if k["name"] == "K2"], j["effective_sclk_mhz"])
PY
}
rocm-smi --showperflevel --showclocks 2>&1 | grep -i "level\|fclk\|mclk\|sclk" | head -n 6
one auto
rocm-smi --setperflevel high 2>&1 | tail -n 2
rocm-smi --showperflevel --showclocks 2>&1 | grep -i "level\|fclk\|mclk\|sclk" | head -n 6
one high
rocm-smi --setperflevel auto 2>&1 | tail -n 1
one auto_again
