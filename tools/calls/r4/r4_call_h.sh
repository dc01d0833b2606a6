#!/bin/bash
# round 4: A/B on one box — k4_gather batch split / workgroup size variants (SG_LIB), hub rows whole in the row sort (SG_ABLATE=0x4000)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/ch_$tag.json 2> $O/ch_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/ch_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], k)
PY
}
V=$PWD/alaz_amd/lib/variants
run base A=1
run hubwhole SG_ABLATE=0x4000
run g_s1r8 SG_LIB=$V/libsg_g_s1r8.so
run g_s2r8 SG_LIB=$V/libsg_g_s2r8.so
run g_s1r4 SG_LIB=$V/libsg_g_s1r4.so
run g_s2r4 SG_LIB=$V/libsg_g_s2r4.so
run g_s4r4 SG_LIB=$V/libsg_g_s4r4.so
run base2 A=1
