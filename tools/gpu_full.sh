#!/bin/bash
# full evidence run: parity tests, bench (C2 + C3), rocprofv3 kernel stats of the timed region, PMC passes.
# TAG=r01k  [SKIP_TESTS=1] [SKIP_C3=1] [SKIP_PMC=1]
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
TAG=${TAG:-r01x}
R="$GRAFT_REPO_ROOT"
if [ -z "$SKIP_TESTS" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
  tail -n 8 gpurun_out/pytest_gpu.log
fi
timeout 600 python bench.py --steps 200 --warmup 20 > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "bench rc=$?"
tail -n 2 gpurun_out/${TAG}_bench_c2.json; tail -n 3 gpurun_out/${TAG}_bench_c2.err
if [ -n "$EXTRA_ENV" ]; then
  for kv in $EXTRA_ENV; do
    echo "== $kv" >> gpurun_out/${TAG}_variants.log
    env $kv timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --overlap-windows 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['kernel_group_us'], 'ms_per_step', round(d['ms_per_step'],4))" >> gpurun_out/${TAG}_variants.log 2>&1
  done
  cat gpurun_out/${TAG}_variants.log
fi
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_${TAG}_c2" -o "$TAG" -- python "$R/bench.py" --steps 200 --warmup 20 --profile-mode > "$R/gpurun_out/${TAG}_rocprof_c2.log" 2>&1
cd "$R" && python tools/rocpd_stats.py gpurun_out/prof_${TAG}_c2/${TAG}_results.db gpurun_out/${TAG}_kernel_stats_c2.txt | head -20
tail -n 1 gpurun_out/${TAG}_rocprof_c2.log | cut -c1-600
if [ -z "$SKIP_C3" ]; then
  timeout 900 python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline --overlap-windows 0 > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; echo "bench c3 rc=$?"
  tail -n 1 gpurun_out/${TAG}_bench_c3.json | cut -c1-1500
  cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/prof_${TAG}_c3" -o "$TAG" -- python "$R/bench.py" --config 3 --steps 10 --warmup 2 --profile-mode > "$R/gpurun_out/${TAG}_rocprof_c3.log" 2>&1
  cd "$R" && python tools/rocpd_stats.py gpurun_out/prof_${TAG}_c3/${TAG}_results.db gpurun_out/${TAG}_kernel_stats_c3.txt | head -20
fi
if [ -z "$SKIP_PMC" ]; then
  CONFIG=2 TAG=$TAG bash tools/gpu_pmc.sh | tail -n 60
fi
