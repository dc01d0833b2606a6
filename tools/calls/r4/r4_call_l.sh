#!/bin/bash
# round 4: row degrees / in-row ranks by LDS histograms (k2_deg_hist) instead of one returning device atomic per edge — parity subset, then A/B on one box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3_full_size_row or edge_cases or empty_and_tiny or capacity_overflow or random_small or row_sort_by_blocks or logical_shards" > $O/cl_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|Extension modules\|amdgpu.ids" $O/cl_pytest.log | tail -n 25
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cl_$tag.json 2> $O/cl_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cl_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run dh128 A=1
run old SG_DH_G=0
run dh64 SG_DH_G=64
run dh128b A=1
run old2 SG_DH_G=0
