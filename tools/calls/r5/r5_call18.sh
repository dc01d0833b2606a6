#!/bin/bash
# round 5, call 18: the C11 client with its ABI-5 part (warm windows by name)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_abi_client.py -m gpu -x -q 2>&1 | grep -v amdgpu.ids | tail -n 12
