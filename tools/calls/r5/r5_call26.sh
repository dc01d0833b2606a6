#!/bin/bash
# round 5, call 26: what is left of the budget on more of the host-fed -m gpu tests against the last library (sg_ingest in pieces)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 78 python -m pytest tests/test_gpu_warm.py tests/test_gpu_parity.py -x -v -m gpu -p no:cacheprovider -k "warm or random_small or windows_in_flight or alive or edge_cases or cpp_graphds or empty_and_tiny or table_updates or capacity_overflow_is" > gpurun_out/r05_call26_pytest.log 2>&1
echo "rc=$?"; grep -c PASSED gpurun_out/r05_call26_pytest.log; grep -v PASSED gpurun_out/r05_call26_pytest.log | tail -n 6
