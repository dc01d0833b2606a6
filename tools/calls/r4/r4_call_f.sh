#!/bin/bash
# round 4: K5 variants A/B on one box (SG_LIB selects the engine library)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for v in v0 v1 v2 v4 v0 v2; do
SG_LIB=$PWD/alaz_amd/lib/variants/libsg_$v.so timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cf_$v.json 2> $O/cf_$v.err; echo "$v rc=$?"; python - $v <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cf_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], 'K5', k.get('K5'), 'K1a', k.get('K1a'), 'K1b', k.get('K1b'), 'K2', k.get('K2'))
PY
done
