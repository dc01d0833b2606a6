#!/bin/bash
# round 4: k2_deg_hist only above 2^19 edges of capacity + kc_prepare's label count loaded with the statistics: parity subset (with the
# forced-histogram test), C2 / C3 lines without the CPU legs, kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
timeout 600 python -m pytest tests -m gpu -q -x -k "lds_histograms or config2 or edge_cases or empty_and_tiny or capacity_overflow or random_small or logical_shards or alive or rccl_entry" > $O/cw_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|Extension modules\|amdgpu.ids" $O/cw_pytest.log | tail -n 6; lap tests
run() { tag=$1; shift; timeout 300 python bench.py --no-cpu-baseline --no-end-to-end "$@" > $O/cw_$tag.json 2> $O/cw_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cw_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run c3 --steps 40; lap c3
run c2 --config 2; lap c2
tools/gpu.sh prof:cw:3 | grep "kc_prep\|k2_\|k1"; tools/gpu.sh prof:cw:2 | grep "kc_prep\|k2_\|k1"; lap prof
