#!/bin/bash
# round 6 (late): where the HIP runtime keeps kernel arguments — HIP_FORCE_DEV_KERNARG=0 / 1 / unset (the runtime's default), one box, the development
# build, two alternating repetitions; every launch of the window reads its ~1 KiB Dev argument before anything else
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2; do
  for v in unset 0 1; do
    if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
    timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed "s/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[HIP_FORCE_DEV_KERNARG $v] /" | cut -c1-210 | tee -a $O/r06_kernarg_ab.txt
  done
done
