#!/bin/bash
# first GPU trip: parity, smoke, probe, bench, rocprof kernel trace
set -x
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import torch; print(torch.__version__, torch.cuda.is_available(), torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
timeout 300 python __graft_entry__.py smoke > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
timeout 120 ./tools/atomic_probe > gpurun_out/atomic_probe.log 2>&1
timeout 600 python bench.py --steps 100 --warmup 10 > gpurun_out/bench.log 2>&1; echo "bench rc=$?" >> gpurun_out/bench.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r1" -o r1 -- python "$GRAFT_REPO_ROOT/bench.py" --steps 50 --warmup 5 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/rocprof.log" 2>&1
cd "$GRAFT_REPO_ROOT"; ls -R gpurun_out | head -50
tail -5 gpurun_out/smoke.log gpurun_out/pytest_gpu.log gpurun_out/bench.log
