#!/bin/bash
# round 6 (late): k5_edge_score with fewer instructions per edge — ReLU as one v_max_f32, the sigmoid's division as v_rcp_f32 (1 ulp; the score's
# tolerance is 1e-5), ref_of_dense as selects — against the development build of commit c38d10a (alaz_amd/lib/ab_fin_dev.so), one box, three
# alternating repetitions; then the parity tests that compare scores with the oracle
# (slower: 59.9-61.5 vs 58.2-58.4 us; the edit is not in the tree)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
for rep in 1 2 3; do
  SG_LIB_DEV=$PWD/alaz_amd/lib/ab_fin_dev.so timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[c38d10a] /' | cut -c1-200 | tee -a $O/r06_k5diet_ab.txt
  timeout 600 python tools/k1_sweep.py 3 "SG_ABLATE=0" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //; s/^/[fewer instructions] /' | cut -c1-200 | tee -a $O/r06_k5diet_ab.txt
done
tools/gpu.sh "tests:config,or,smoke,or,golden,or,variant" | tail -n 8
tools/gpu.sh smoke
