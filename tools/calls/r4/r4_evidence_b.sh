#!/bin/bash
# round-4 evidence, second call: config 2, the sharded path at world = 1 (weak + strong, rows verified), config 5 as one shard of eight
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh bench:r04_z_c2:--config,2,--cpu-seconds,3 prof:r04_z:2 pmc:r04_z:2
for sc in weak strong; do
  SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --scaling $sc --verify > $O/r04_z_sharded1_$sc.json 2> $O/r04_z_sharded1_$sc.err
  echo "sharded1 $sc rc=$?"; tail -n 1 $O/r04_z_sharded1_$sc.json | cut -c1-400
done
timeout 900 python bench.py --config 5 --shard-of 8 --no-cpu-baseline --no-end-to-end > $O/r04_z_c5_shard.json 2> $O/r04_z_c5_shard.err; echo "c5 shard rc=$?"; tail -n 1 $O/r04_z_c5_shard.json | cut -c1-300
timeout 600 python tools/c5_stream.py --shard-of 8 --windows 6 --expand 4000000 > $O/r04_z_c5_stream_shard.json 2> $O/r04_z_c5_stream_shard.err; echo "c5 stream shard rc=$?"; cut -c1-1200 $O/r04_z_c5_stream_shard.json; tail -n 2 $O/r04_z_c5_stream_shard.err
SG_BENCH_E2E=pinned timeout 600 python bench.py --no-cpu-baseline --overlap-windows 0 --steps 5 > $O/r04_z_e2e_pinned_again.json 2> /dev/null; python -c "
import json; j=json.loads(open('$O/r04_z_e2e_pinned_again.json').read().strip().splitlines()[-1]); r=j['end_to_end']['registered_memory']; print('pinned again', r['events_per_s'], r['frac_of_pcie_bound'], r['ms_per_window'])"
