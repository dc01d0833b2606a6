#!/bin/bash
# round 6, call 18: behind the header split (no code change): the whole -m gpu suite, the PMC passes on exactly these sources, the C3 line's GPU legs
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/gpu.sh bench:r06_final_c3_gpu_legs:--steps,20,--warmup,5,--no-cpu-baseline,--no-end-to-end | cut -c1-400
tools/gpu.sh pmc:r06_final:3 | tail -n 12
tools/gpu.sh prof:r06_final:3 | head -n 22
tools/gpu.sh smoke
tools/gpu.sh tests | tail -n 6
