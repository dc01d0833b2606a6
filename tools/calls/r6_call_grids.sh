#!/bin/bash
# round 6 (late): whole rounds of workgroups.  k5_edge_score runs 3 waves per SIMD (144 registers) = 768 workgroups of 256 threads at a time; its
# grid of 2 048 is 2.67 rounds.  k3_node_features' edge workgroups: 1 280 at a time (90 registers), 4 096 launched.  One box, the development
# build (SG_K5_GRID / SG_K3_EGRID cap the grids), two repetitions
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
V="SG_ABLATE=0;SG_K5_GRID=768;SG_K5_GRID=1536;SG_K5_GRID=2304;SG_K5_GRID=1024;SG_K3_EGRID=1280;SG_K3_EGRID=2560"
IFS=';' read -ra A <<< "$V"
timeout 1200 python tools/k1_sweep.py 3 "${A[@]}" "${A[@]}" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //' | cut -c1-200 | tee $O/r06_grids_ab.txt
