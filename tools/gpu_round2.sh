#!/bin/bash
# Round-2 measurement set on one MI355X box: bench lines (C3 default, C2), rocprofv3 kernel stats of the timed region,
# PMC passes (FETCH_SIZE, WRITE_SIZE, separate runs) for the K1 kernels at C3 and C2.  TAG names the files.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
TAG=${TAG:-r02_z}
timeout 900 python bench.py --cpu-seconds ${CPU_SECONDS:-6} > gpurun_out/${TAG}_bench_c3.json 2> gpurun_out/${TAG}_bench_c3.err; echo "c3 rc=$?"; cut -c1-700 gpurun_out/${TAG}_bench_c3.json
timeout 600 python bench.py --config 2 --cpu-seconds ${CPU_SECONDS:-6} > gpurun_out/${TAG}_bench_c2.json 2> gpurun_out/${TAG}_bench_c2.err; echo "c2 rc=$?"; cut -c1-500 gpurun_out/${TAG}_bench_c2.json
cd /tmp
for C in 3 2; do
  timeout 600 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_c$C" -o kt -- python "$GRAFT_REPO_ROOT/bench.py" --config $C --profile-mode > "$GRAFT_REPO_ROOT/gpurun_out/rocprof_${TAG}_c$C.log" 2>&1
  python "$GRAFT_REPO_ROOT/tools/rocpd_stats.py" "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_c$C/kt_results.db" "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_kernel_stats_c$C.txt" | head -16
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pmc_${c}_c$C"
    timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc_${c}_c$C" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --config $C --steps 10 --warmup 2 --profile-mode > "$GRAFT_REPO_ROOT/gpurun_out/pmc_${c}_c$C.log" 2>&1
  done
  python "$GRAFT_REPO_ROOT/tools/pmc_k1_json.py" "$GRAFT_REPO_ROOT/gpurun_out/pmc_FETCH_SIZE_c$C" "$GRAFT_REPO_ROOT/gpurun_out/pmc_WRITE_SIZE_c$C" $C $TAG > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_k1_c$C.json"
  python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$GRAFT_REPO_ROOT/gpurun_out/pmc_FETCH_SIZE_c$C" "$GRAFT_REPO_ROOT/gpurun_out/pmc_WRITE_SIZE_c$C" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_summary_c$C.txt"
  cat "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_pmc_k1_c$C.json"
  rm -rf "$GRAFT_REPO_ROOT/gpurun_out/prof_${TAG}_c$C"
done
# SQ / LDS activity of the K1 kernels at C3 (why they are where they are: issue, wait and LDS-array cycles)
i=0
for c in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES"; do
  i=$((i+1)); rm -rf "$GRAFT_REPO_ROOT/gpurun_out/pmcq_$i"
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmcq_$i" -o pmc -- python "$GRAFT_REPO_ROOT/bench.py" --config 3 --steps 6 --warmup 2 --profile-mode > "$GRAFT_REPO_ROOT/gpurun_out/pmcq_$i.log" 2>&1
done
python "$GRAFT_REPO_ROOT/tools/pmc_summary.py" "$GRAFT_REPO_ROOT/gpurun_out/pmcq_1" "$GRAFT_REPO_ROOT/gpurun_out/pmcq_2" | grep -E "k1a|k1b|==" > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_k1_sq_counters_c3.txt"
cat "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_k1_sq_counters_c3.txt"
# config 5 as a stream of raw records (nominal rate, then as fast as the host side can offer)
cd "$GRAFT_REPO_ROOT"
timeout 300 python tools/c5_stream.py > gpurun_out/${TAG}_c5_stream_5M.json 2> gpurun_out/${TAG}_c5_stream_5M.err; cut -c1-900 gpurun_out/${TAG}_c5_stream_5M.json
timeout 300 python tools/c5_stream.py --rate 3e7 --windows 5 --feeders 16 > gpurun_out/${TAG}_c5_stream_max.json 2> gpurun_out/${TAG}_c5_stream_max.err; cut -c1-900 gpurun_out/${TAG}_c5_stream_max.json
