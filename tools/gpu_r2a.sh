#!/bin/bash
# round 2, first GPU pass on the rewritten K1: parity tests, then C2 / C3 bench lines with phase stamps
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 30 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --config 2 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/bench_c2.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c2.log
tail -n 3 gpurun_out/bench_c2.log | cut -c1-1500
timeout 600 python bench.py --config 3 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/bench_c3.log 2>&1; echo "rc=$?" >> gpurun_out/bench_c3.log
tail -n 3 gpurun_out/bench_c3.log | cut -c1-1500
SG_ABLATE=0x100 timeout 300 python tools/stamps.py 2 > gpurun_out/stamps_c2.log 2>&1; tail -n 30 gpurun_out/stamps_c2.log
SG_ABLATE=0x100 timeout 600 python tools/stamps.py 3 > gpurun_out/stamps_c3.log 2>&1; tail -n 30 gpurun_out/stamps_c3.log
