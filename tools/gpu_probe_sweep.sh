#!/bin/bash
# per-shard kernel time of the weak-scaling workload at world = 1, 2, 4, 8 (logical shards on one GPU, one stream)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
for w in 1 2 4 8; do
  cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_probe_w$w" -o p -- python "$GRAFT_REPO_ROOT/tools/shard_scale_probe.py" $w 10 ${GRAPH:-fixed} > "$GRAFT_REPO_ROOT/gpurun_out/probe_w$w.log" 2>&1
  cd "$GRAFT_REPO_ROOT" && python tools/rocpd_stats.py gpurun_out/prof_probe_w$w/p_results.db gpurun_out/probe_w${w}_kernel_stats.txt > /dev/null
  python - "$w" <<'PY'
import sys
w = int(sys.argv[1]); tot = 0.0; rows = []
for ln in open(f"gpurun_out/probe_w{w}_kernel_stats.txt").read().splitlines()[1:]:
    name = ln[:64].strip(); f = ln[64:].split()
    if not f or any(x in name for x in ("at::native", "rocclr", "rccl")): continue
    calls, avg = int(f[0]), float(f[2])
    per_window = avg * calls / (13 * w)          # 3 warm-up + 10 timed windows per shard
    tot += per_window; rows.append((name[:28], round(per_window, 1)))
print(f"world {w}: engine kernels per shard-window = {tot:.0f} us ", sorted(rows, key=lambda r: -r[1])[:8])
PY
done
