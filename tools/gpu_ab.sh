#!/bin/bash
# same-box A/B of engine builds: LIBS="name=path ..." (paths relative to the repo), REPS=2
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
TAG=${TAG:-ab}; REPS=${REPS:-2}
: > gpurun_out/${TAG}_ab.log
for r in $(seq 1 $REPS); do
  for kv in $LIBS; do
    name=${kv%%=*}; path=${kv#*=}
    SG_LIB_PATH="$GRAFT_REPO_ROOT/$path" timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --overlap-windows 0 ${BENCH_ARGS} 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', d['kernel_group_us'], 'ms_per_step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],4))" >> gpurun_out/${TAG}_ab.log 2>&1
  done
done
cat gpurun_out/${TAG}_ab.log
