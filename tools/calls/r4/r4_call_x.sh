#!/bin/bash
# round 4: pass B with count + duration sum in one 64-bit LDS add (k1b_stream_merge<.., PACK>): parity subset, A/B against SG_K1B_PACK=0 at C3 and C2
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 400 python -m pytest tests -m gpu -q -x -k "pass_b or lds_histograms or config2 or edge_cases or empty_and_tiny or capacity_overflow or random_small or logical_shards or alive or windows_in_flight" > $O/cx_pytest.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|Extension modules\|amdgpu.ids" $O/cx_pytest.log | tail -n 4
tools/gpu.sh "sweep:3:;SG_K1B_PACK=0;;SG_K1B_PACK=0" "sweep:2:;SG_K1B_PACK=0;;SG_K1B_PACK=0"
