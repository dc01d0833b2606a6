#!/bin/bash
# round 4: pass B's deg atomics back to back — parity subset, default bench twice
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "config2_full or config3_full_size_row or edge_cases or overflow or tiny or logical_shards or k4_gather or second_long" > $O/ci_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/ci_pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/ci_$tag.json 2> $O/ci_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/ci_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run base A=1
run nodeg SG_ABLATE=0x20
run base2 A=1
