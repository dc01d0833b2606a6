#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu.log
tail -n 25 gpurun_out/pytest_gpu.log
timeout 600 python tools/k1_sweep.py 3 "" "SG_K3_SLICES=16" "SG_K3_SLICES=64" > gpurun_out/sweep_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c3.log
timeout 300 python tools/k1_sweep.py 2 "" > gpurun_out/sweep_c2.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep_c2.log
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_r2h" -o r2h -- python "$GRAFT_REPO_ROOT/bench.py" --steps 10 --warmup 2 --profile-mode > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_r2h.log" 2>&1
cd "$GRAFT_REPO_ROOT" && python tools/rocpd_stats.py gpurun_out/prof_r2h/r2h_results.db gpurun_out/r2h_kernel_stats_c3.txt | head -24
