#!/bin/bash
# round 5, call 6: BASELINE config 2 (119 us in round 2, 147 in round 4) — which pass A, warm windows or not, on one box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
run() { # tag, env...
  tag=$1; shift
  env "$@" python bench.py --config 2 --no-cpu-baseline --no-end-to-end > $O/r05_f_${tag}_c2_bench.json 2> $O/r05_f_${tag}.err
  python - "$tag" <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r05_f_{sys.argv[1]}_c2_bench.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(j["ms_per_step"] * 1e3, 1), "us  median", j["per_step"]["median_ms"], "K1 frac", round(j["roofline"]["frac"], 4), [(k["name"], k["us_per_window"]) for k in j["kernels"]], j["roofline"]["geometry"]["pass_a_teams"], j["warm_windows"]["engine_keeps_state"])
PY
}
run default X=1; lap a
run nowarm SG_WARM=0; lap b
run tile SG_K1A=tile; lap c
run tile_nowarm SG_K1A=tile SG_WARM=0; lap d
run legacy SG_K1_LEGACY=1; lap e
SG_WARM=1 tools/gpu.sh prof:r05_f_default:2 | head -n 22; lap prof
SG_K1A=tile SG_WARM=0 tools/gpu.sh prof:r05_f_tile_nowarm:2 | head -n 18; lap prof2
