#!/bin/bash
# round 4: small device arrays carved out of 256 MiB chunks (SG_ARENA, default on) against one hipMalloc per array — A/B on one box, kernel stats both ways
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/cs_$tag.json 2> $O/cs_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/cs_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run arena A=1
run malloc SG_ARENA=0
run arena2 A=1
run malloc2 SG_ARENA=0
bash tools/gpu.sh prof:cs:3 | head -24
SG_ARENA=0 bash tools/gpu.sh prof:cs0:3 | head -24
bash tools/box_probe.sh cs 2>/dev/null | grep -i "vm_\|fragment\|uptime" | head
