#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out; rm -f gpurun_out/np.log
for np in 128 256 512 1024; do
  echo "SG_NP=$np" >> gpurun_out/np.log
  SG_NP=$np timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_group_us'], 'ms_per_step', round(d['ms_per_step'],4), 'E', d['config']['edges_per_window'])" >> gpurun_out/np.log 2>&1
done
cat gpurun_out/np.log
