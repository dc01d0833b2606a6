#!/bin/bash
# round 4: dependent round trips trimmed in kc_prepare, k3_in_reduce, k2_deg_hist, k2_rowptr, k2_scatter_parts — parity, bench, kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "config1 or config2 or config3 or edge_cases or empty_and_tiny or random_small or row_sort_by_blocks or logical_shards or histogram or config5_mixed or alive" > $O/cr_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|Extension modules\|amdgpu.ids" $O/cr_pytest.log | tail -n 25
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cr_$tag.json 2> $O/cr_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cr_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run a A=1
run b A=1
bash tools/gpu.sh prof:cr:3 | head -24
