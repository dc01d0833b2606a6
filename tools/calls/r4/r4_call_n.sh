#!/bin/bash
# round 4: k2_rowptr<64 rows, DH> (236 workgroups pull the count table) — parity subset, A/B on one box, kernel stats
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3_full_size_row or edge_cases or empty_and_tiny or capacity_overflow or random_small or row_sort_by_blocks or logical_shards or histogram" > $O/co_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|Extension modules\|amdgpu.ids" $O/co_pytest.log | tail -n 25
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/co_$tag.json 2> $O/co_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/co_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run dh A=1
run old SG_DH_G=0
run dh32 SG_DH_G=32
run dh2 A=1
bash tools/gpu.sh prof:co:3 | head -24
