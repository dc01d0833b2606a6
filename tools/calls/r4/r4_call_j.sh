#!/bin/bash
# round 4: pass B compaction old vs new on one box; what the box is under load
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cj_$tag.json 2> $O/cj_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cj_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
V=$PWD/alaz_amd/lib/variants
run new A=1
run old SG_LIB=$V/libsg_olddeg.so
run new2 A=1
run old2 SG_LIB=$V/libsg_olddeg.so
bash tools/box_probe.sh cj
