#!/bin/bash
# round 4: in-statistics slice count (env), then what the box is under load
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --steps 40 > $O/ck_$tag.json 2> $O/ck_$tag.err; echo "$tag rc=$?"; python - $tag <<'PY'
import json,sys
j=json.loads(open('gpurun_out/ck_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
k={x['name']:x['us_per_window'] for x in j.get('kernels',[])}
print(sys.argv[1], j['ms_per_step'], j['roofline']['frac'], k)
PY
}
run s32 A=1
run s48 SG_K3_SLICES=48
run s51 SG_K3_SLICES=51
run s64 SG_K3_SLICES=64
run s16 SG_K3_SLICES=16
bash tools/box_probe.sh ck | grep -v "card[0-9]*/device/\(mem_info_vram_total\|current_\)"
