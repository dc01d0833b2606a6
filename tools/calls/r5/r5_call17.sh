#!/bin/bash
# round 5, call 17: the final tree — smoke() and the whole -m gpu suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -n 2
tools/gpu.sh tests | tail -n 6
