#!/bin/bash
# round 5, call 13: the sharded bench at world = 1 after the settle-loop refactor (strong default + weak, rows verified)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --verify > $O/r05_zz_sharded1.json 2> $O/r05_zz_sharded1.err; echo "rc=$?"
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r05_zz_sharded1.json").read().strip().splitlines()[-1])
print({k: j.get(k) for k in ("value", "ms_per_step", "scaling", "comm_us_per_window", "rows_verified", "weak", "error")})
PY
tail -n 3 $O/r05_zz_sharded1.err
