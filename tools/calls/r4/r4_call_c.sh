#!/bin/bash
# round 4: pass A's first chunk rotates per launch — parity subset, the C5 shard stream (drops?), the C5 shard window, the default bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 900 python -m pytest tests -m gpu -x -q -k "edge_cases or stream or batches or config1 or config2 or sweep or small or churn or pinned" > $O/cc_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/cc_pytest.log
timeout 600 python tools/c5_stream.py --shard-of 8 --windows 6 --expand 4000000 > $O/cc_c5_stream_shard.json 2> $O/cc_c5_stream_shard.err; echo "c5 stream shard rc=$?"; cut -c1-1500 $O/cc_c5_stream_shard.json; tail -n 2 $O/cc_c5_stream_shard.err
timeout 900 python bench.py --config 5 --shard-of 8 --no-cpu-baseline --no-end-to-end > $O/cc_c5_shard.json 2> $O/cc_c5_shard.err; echo "c5 shard rc=$?"; tail -n 1 $O/cc_c5_shard.json | cut -c1-600
timeout 600 python bench.py --no-cpu-baseline --no-end-to-end > $O/cc_c3.json 2> $O/cc_c3.err; echo "c3 rc=$?"; tail -n 1 $O/cc_c3.json | cut -c1-700
