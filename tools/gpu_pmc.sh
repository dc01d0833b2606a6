#!/bin/bash
# HBM traffic of the K1 kernels: separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --steps 20 --warmup 3 --no-cpu-baseline > "$GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log" 2>&1
done
cd "$GRAFT_REPO_ROOT"
ls -R gpurun_out/pmc_FETCH_SIZE | head; python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE | tee gpurun_out/pmc_summary.txt
