#!/bin/bash
# (the kernel this call measured — k4_gather_coop behind SG_K4_COOP / SG_K4_COOP_MIN / SG_K4_EDGE_RIDE — was slower and is not in the tree: DESIGN.md §3 K4, profiles/r06_k4coop_ab_c3.txt)
# round 6: K4's gather with the long chains (hub blocks, rows of more than SG_K4_COOP_MIN edges) taken by whole workgroups, split by slot
# (k4_gather_coop), and the edge features riding in the first layer's gather launch — against k4_gather, one box, the development build;
# the bitwise test first, then the whole -m gpu suite with both switched on
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
timeout 600 python -m pytest tests -m gpu -q -x -k "k4_gather_by_whole or k4_gather_launch" 2>&1 | grep -v amdgpu.ids | tail -n 15 | tee $O/r06_k4coop_test.txt
V0="SG_ABLATE=0"; V1="SG_K4_COOP=1"; V2="SG_K4_COOP=1 SG_K4_EDGE_RIDE=1"; V3="SG_K4_COOP=1 SG_K4_COOP_MIN=128 SG_K4_EDGE_RIDE=1"; V4="SG_K4_COOP=1 SG_K4_COOP_MIN=32 SG_K4_EDGE_RIDE=1"
timeout 900 python tools/k1_sweep.py 3 "$V0" "$V1" "$V2" "$V3" "$V4" "$V0" "$V1" "$V2" "$V3" "$V4" 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee $O/r06_k4coop_ab.txt
SG_K4_COOP=1 SG_K4_EDGE_RIDE=1 tools/gpu.sh tests | tail -n 12
