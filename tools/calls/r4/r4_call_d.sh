#!/bin/bash
# round 4: the row sort at large node counts (wave rank sort, LDS arrays sized by the node capacity) — parity at config 5 / 3 / 2, the C5 shard window, the default bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "config5 or logical_shards or row_sort or config3_full_size_row or config2_full or histogram_engine or edge_cases" > $O/cd_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/cd_pytest.log
timeout 900 python bench.py --config 5 --shard-of 8 --no-cpu-baseline --no-end-to-end > $O/cd_c5_shard.json 2> $O/cd_c5_shard.err; echo "c5 shard rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/cd_c5_shard.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['value']); [print(k) for k in j.get('kernels',[])]
PY
timeout 600 python bench.py --no-cpu-baseline --no-end-to-end > $O/cd_c3.json 2> $O/cd_c3.err; echo "c3 rc=$?"; python - <<'PY'
import json
j=json.loads(open('gpurun_out/cd_c3.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['pass_a_us'], j['roofline']['pass_b_us']); [print(k) for k in j.get('kernels',[])]
PY
