#!/bin/bash
# round 4: k2_deg_hist with 16 loads in flight; pass B with the batched merge (SG_K1B_BM=1) — parity subset under BM, then A/B on one box
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
SG_K1B_BM=1 timeout 900 python -m pytest tests -m gpu -q -x -k "config2 or config3_full_size_row or edge_cases or empty_and_tiny or capacity_overflow or random_small or pass_b_second or overflow_paths" > $O/cm_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|Extension modules\|amdgpu.ids" $O/cm_pytest.log | tail -n 25
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cm_$tag.json 2> $O/cm_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cm_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run dh A=1
run bm SG_K1B_BM=1
run old SG_DH_G=0
run dh2 A=1
run bm2 SG_K1B_BM=1
