#!/bin/bash
# round 4: A/B on one box — the shipped build, pass A without its edge cache (SG_ABLATE=8), the row sort at four waves per SIMD
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cg_$tag.json 2> $O/cg_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cg_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], k, 'dropped', j['config'].get('events_dropped_cap'))
PY
}
run base A=1
run nocache SG_ABLATE=8
run r128 SG_LIB=$PWD/alaz_amd/lib/variants/libsg_r128.so
run base2 A=1
run nocache2 SG_ABLATE=8
