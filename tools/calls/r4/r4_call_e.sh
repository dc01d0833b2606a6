#!/bin/bash
# round 4: K5 with two iterations in flight — parity subset, default bench
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 1200 python -m pytest tests -m gpu -x -q -k "config2_full or config3_full_size_row or edge_cases or histogram or capacity_overflow or tiny or logical_shards" > $O/ce_pytest.log 2>&1; echo "pytest rc=$?"; tail -n 3 $O/ce_pytest.log
for i in 1 2; do
timeout 600 python bench.py --no-cpu-baseline --no-end-to-end > $O/ce_c3_$i.json 2> $O/ce_c3_$i.err; echo "c3 rc=$?"; python - $i <<'PY'
import json,sys
j=json.loads(open('gpurun_out/ce_c3_%s.json'%sys.argv[1]).read().strip().splitlines()[-1])
print(j['ms_per_step'], j['value'], j['roofline']['frac'], j['roofline']['pass_a_us'], j['roofline']['pass_b_us']); [print(k) for k in j.get('kernels',[])]
PY
done
