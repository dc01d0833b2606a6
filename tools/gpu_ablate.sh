#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for ab in 0 2 1 5 4 6; do
  echo "SG_ABLATE=$ab" >> gpurun_out/ablate.log
  SG_ABLATE=$ab timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_group_us'], 'ms_per_step', round(d['ms_per_step'],4))" >> gpurun_out/ablate.log 2>&1
done
cat gpurun_out/ablate.log
