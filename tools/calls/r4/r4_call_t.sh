#!/bin/bash
# round 4, final evidence at the head of the kernels (c3e1ebe): full -m gpu suite, C3 bench line (with the CPU baseline and end to end),
# rocprofv3 kernel stats + FETCH/WRITE PMC of the same command, C2 line + stats, the sharded path at world = 1 with the rows verified
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
tools/gpu.sh box:r04_zz; lap box
tools/gpu.sh tests | tail -n 6; lap tests
tools/gpu.sh smoke; lap smoke
tools/gpu.sh bench:r04_zz_c3:--cpu-seconds,4 | cut -c1-600; lap bench3
tools/gpu.sh prof:r04_zz:3; lap prof3
tools/gpu.sh pmc:r04_zz:3 | tail -n 12; lap pmc3
tools/gpu.sh bench:r04_zz_c2:--config,2,--cpu-seconds,2 | cut -c1-300; lap bench2
tools/gpu.sh prof:r04_zz:2 | head -n 16; lap prof2
SG_FORCE_SHARDED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --scaling strong --verify > $O/r04_zz_sharded1_strong.json 2> $O/r04_zz_sharded1_strong.err
echo "sharded1 strong rc=$?"; tail -n 1 $O/r04_zz_sharded1_strong.json | cut -c1-400; lap sharded
