#!/bin/bash
# round 6 (late): k4_gather as resident workgroups striding over the tiles (4 per CU at its 4 waves per SIMD) instead of one workgroup per tile
# of four rows (3 766 at C3); one box, the development build (SG_K4G_GRID caps the grid), two repetitions
# (SG_K4G_GRID was a knob of the development build for this call only; slower, not in the tree)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
V="SG_ABLATE=0;SG_K4G_GRID=1024;SG_K4G_GRID=2048;SG_K4G_GRID=1536;SG_K4G_GRID=512"
IFS=';' read -ra A <<< "$V"
timeout 1200 python tools/k1_sweep.py 3 "${A[@]}" "${A[@]}" 2>&1 | grep -v amdgpu.ids | sed 's/narrow np 512 x2 ht 2048 ct 1024 l2lds 2 | //' | cut -c1-200 | tee $O/r06_grids2_ab.txt
