#!/bin/bash
# end-of-round sanity on the GPU box: parity tests, smoke, the default bench line, and the multi-GPU code path
# (torch.distributed launcher + RCCL calls) at world = 1.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 300 python bench.py > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; echo "bench rc=$?"; tail -n 1 gpurun_out/final_bench.json | cut -c1-420
SG_FORCE_SHARDED=1 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 100 --warmup 10 --no-cpu-baseline > gpurun_out/final_sharded1.json 2> gpurun_out/final_sharded1.err; echo "sharded rc=$?"; tail -n 1 gpurun_out/final_sharded1.json | cut -c1-420
