#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; echo "default rc=$?"; tail -c 6000 gpurun_out/bench_default.json; tail -n 5 gpurun_out/bench_default.err
timeout 600 python bench.py --config 2 > gpurun_out/bench_c2.json 2> gpurun_out/bench_c2.err; echo "c2 rc=$?"; tail -c 3500 gpurun_out/bench_c2.json
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
