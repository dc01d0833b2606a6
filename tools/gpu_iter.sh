#!/bin/bash
# standard iteration: parity tests, then bench (+ optional rocprof with TAG)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 25 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --steps 100 --warmup 10 ${BENCH_ARGS} > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
tail -n 4 gpurun_out/bench.log
if [ -n "$TAG" ]; then
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_$TAG" -o "$TAG" -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_$TAG.log" 2>&1
  cd "$GRAFT_REPO_ROOT" && python tools/rocpd_stats.py gpurun_out/prof_$TAG/${TAG}_results.db gpurun_out/${TAG}_kernel_stats.txt | head -30
fi
