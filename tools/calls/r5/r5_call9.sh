#!/bin/bash
# round 5, call 9: end-to-end feed at two staging batch sizes; world-1 sharded path on the new default (strong + weak)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
for mb in 262144 1048576; do
  SG_BENCH_MAX_BATCH=$mb python bench.py --no-cpu-baseline > $O/r05_i_e2e_${mb}.json 2> $O/r05_i_e2e_${mb}.err
  python - $mb <<'PY'
import json, sys
j = json.loads(open(f"gpurun_out/r05_i_e2e_{sys.argv[1]}.json").read().strip().splitlines()[-1])
e = j["end_to_end"]; r = e.get("registered_memory", {})
print(sys.argv[1], "window", round(j["ms_per_step"] * 1e3, 1), "e2e pageable", round(e["events_per_s"] / 1e9, 3), "G ev/s", e["frac_of_pcie_bound"], e["ring_full_retries"], e["pcie_measured_GBs"], "registered", round(r.get("events_per_s", 0) / 1e9, 3), r.get("frac_of_pcie_bound"), "box", j["box"]["hbm_loaded_latency_ns"], j["roofline"]["pass_a_us"])
PY
  lap e2e_$mb
done
SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --verify > $O/r05_i_sharded1.json 2> $O/r05_i_sharded1.err
echo "rc=$?"; tail -n 1 $O/r05_i_sharded1.json | cut -c1-1500; tail -n 3 $O/r05_i_sharded1.err; lap sharded1
