#!/bin/bash
# rocprofv3 kernel stats of a few windows of one config through tools/k1_sweep.py (CONFIG, VAR)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp SWEEP_STEPS=${SWEEP_STEPS:-6}
mkdir -p gpurun_out
cd /tmp
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_iter"
timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_iter" -o kt -- python "$GRAFT_REPO_ROOT/tools/k1_sweep.py" ${CONFIG:-3} "${VAR:-}" > "$GRAFT_REPO_ROOT/gpurun_out/prof_iter.log" 2>&1
python "$GRAFT_REPO_ROOT/tools/rocpd_stats.py" "$GRAFT_REPO_ROOT/gpurun_out/prof_iter/kt_results.db" "$GRAFT_REPO_ROOT/gpurun_out/prof_iter_stats.txt" | head -24
rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_iter"
