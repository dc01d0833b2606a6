#!/bin/bash
# One script for everything that runs on the GPU box.  Usage (through gpurun):  tools/gpu.sh STEP [STEP ...]
# A step is NAME or NAME:ARG[:ARG...] (',' inside an ARG stands for a blank); steps run in order, output under gpurun_out/.
#   tests[:KEXPR]            pytest -m gpu (all tests, no -x), optional -k expression
#   smoke                    __graft_entry__.smoke()
#   sweep:CONFIG:VAR;VAR     tools/k1_sweep.py CONFIG with the ';'-separated environment variants (e.g. sweep:3:SG_NP=1024,SG_HT=2048;SG_K1_LEGACY=1)
#   stamps:CONFIG[:ENV]      phase stamps of the K1 kernels (tools/stamps.py)
#   bench:TAG[:ARGS]         bench.py ARGS -> gpurun_out/TAG_bench.json
#   prof:TAG:CONFIG          rocprofv3 --kernel-trace --stats of bench.py --profile-mode -> TAG_kernel_stats_cCONFIG.txt
#   pmc:TAG:CONFIG           FETCH_SIZE / WRITE_SIZE (separate passes) of the K1 kernels -> TAG_pmc_k1_cCONFIG.json, TAG_pmc_summary_cCONFIG.txt
#   mfma:TAG:CONFIG          SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES pass -> TAG_mfma_cCONFIG.txt
#   sq:TAG:CONFIG            SQ / LDS activity counters of the K1 kernels -> TAG_k1_sq_counters_cCONFIG.txt
#   sharded1:TAG             bench.py under torch.distributed.run at world = 1 with the sharded pipeline forced (RCCL path)
#   probe:TAG:WORLD[:CFG]    rocprofv3 kernel stats of tools/shard_scale_probe.py WORLD 6 fixed CFG (per-shard kernel time of C4)
#   c5stream:TAG[:ARGS]      tools/c5_stream.py ARGS -> TAG_c5_stream.json
#   calib:TAG                tools/pmc_calib under FETCH_SIZE / WRITE_SIZE -> TAG_pmc_calibration.json (factors reported / real bytes)
#   box:TAG                  what this box is: clocks, power cap, memory / compute partition mode -> TAG_box.txt (the pool's boxes differ by up to 30 % on pass A)
#   run:CMD                  any command (',' = blank)
cd "$GRAFT_REPO_ROOT" || exit 1
R="$GRAFT_REPO_ROOT"; O="$R/gpurun_out"
mkdir -p "$O"
export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
HEAD_ID=$(cat "$R/.head_id" 2>/dev/null || echo unknown)
for step in "$@"; do
  IFS=':' read -r name a1 a2 a3 <<< "$step"
  a1=${a1//,/ }; a2=${a2//,/ }; a3=${a3//,/ }
  echo "==== $step"
  case "$name" in
    tests)
      if [ -n "$a1" ]; then timeout 1500 python -m pytest tests -m gpu -q -k "$a1" > "$O/pytest_gpu.log" 2>&1; else timeout 1500 python -m pytest tests -m gpu -q > "$O/pytest_gpu.log" 2>&1; fi
      echo "pytest rc=$?" >> "$O/pytest_gpu.log"; grep -v "^  File\|Extension modules\|amdgpu.ids" "$O/pytest_gpu.log" | tail -n 40 ;;
    smoke) timeout 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 3 ;;
    sweep) IFS=';' read -ra V <<< "$a2"; timeout 900 python tools/k1_sweep.py "$a1" "${V[@]}" 2>&1 | grep -v amdgpu.ids | tee -a "$O/sweep_c$a1.log" ;;
    stamps) env $a2 timeout 300 python tools/stamps.py "$a1" 2>&1 | grep -v amdgpu.ids | tee "$O/stamps_c$a1.log" ;;
    bench) timeout 900 python bench.py $a2 > "$O/${a1}_bench.json" 2> "$O/${a1}_bench.err"; echo "bench rc=$?"; tail -n 1 "$O/${a1}_bench.json" | cut -c1-1500 ;;
    prof)
      ( cd /tmp && rm -rf "$O/prof_$a1" && timeout 600 rocprofv3 --kernel-trace --stats -d "$O/prof_$a1" -o kt -- python "$R/bench.py" --config "$a2" --profile-mode > "$O/rocprof_${a1}_c$a2.log" 2>&1
        python "$R/tools/rocpd_stats.py" "$O/prof_$a1/kt_results.db" "$O/${a1}_kernel_stats_c$a2.txt" | head -n 24; rm -rf "$O/prof_$a1" ) ;;
    pmc)
      ( cd /tmp
        for c in FETCH_SIZE WRITE_SIZE; do
          rm -rf "$O/pmc_${c}_c$a2"
          timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmc_${c}_c$a2" -o pmc -- python "$R/bench.py" --config "$a2" --steps 10 --warmup 2 --profile-mode > "$O/pmc_${c}_c$a2.log" 2>&1
        done
        python "$R/tools/pmc_k1_json.py" "$O/pmc_FETCH_SIZE_c$a2" "$O/pmc_WRITE_SIZE_c$a2" "$a2" "$a1" "$HEAD_ID" > "$O/${a1}_pmc_k1_c$a2.json"
        python "$R/tools/pmc_summary.py" "$O/pmc_FETCH_SIZE_c$a2" "$O/pmc_WRITE_SIZE_c$a2" > "$O/${a1}_pmc_summary_c$a2.txt"
        cat "$O/${a1}_pmc_k1_c$a2.json"; rm -rf "$O/pmc_FETCH_SIZE_c$a2" "$O/pmc_WRITE_SIZE_c$a2" ) ;;
    mfma)
      ( cd /tmp; rm -rf "$O/pmc_mfma"
        timeout 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES --output-format csv -d "$O/pmc_mfma" -o pmc -- python "$R/bench.py" --config "$a2" --steps 10 --warmup 2 --profile-mode > "$O/pmc_mfma.log" 2>&1
        { echo "# head $HEAD_ID, bench.py --config $a2 --profile-mode, per launch"; python "$R/tools/pmc_summary.py" "$O/pmc_mfma"; } > "$O/${a1}_mfma_c$a2.txt"
        grep -E "k4_|==|#" "$O/${a1}_mfma_c$a2.txt"; rm -rf "$O/pmc_mfma" ) ;;
    sq)
      ( cd /tmp; i=0
        for c in "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_LDS_ATOMIC SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_WAVES" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_HIT_sum TCC_MISS_sum"; do
          i=$((i+1)); rm -rf "$O/pmcq_$i"
          timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/pmcq_$i" -o pmc -- python "$R/bench.py" --config "$a2" --steps 6 --warmup 2 --profile-mode > "$O/pmcq_$i.log" 2>&1
        done
        { echo "# head $HEAD_ID, bench.py --config $a2 --profile-mode, per launch"; python "$R/tools/pmc_summary.py" "$O/pmcq_1" "$O/pmcq_2" "$O/pmcq_3" | grep -E "k1a|k1b|=="; } > "$O/${a1}_k1_sq_counters_c$a2.txt"
        cat "$O/${a1}_k1_sq_counters_c$a2.txt"; rm -rf "$O/pmcq_1" "$O/pmcq_2" "$O/pmcq_3" ) ;;
    sharded1)
      SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline > "$O/${a1}_sharded1.json" 2> "$O/${a1}_sharded1.err"
      echo "rc=$?"; tail -n 1 "$O/${a1}_sharded1.json" | cut -c1-900 ;;
    probe)   # per-shard kernel time of the WORLD-shard workload of config a3 (default 3 = C4) from the profiler, not from event pairs
      ( cd /tmp && rm -rf "$O/prof_probe" && timeout 900 rocprofv3 --kernel-trace --stats -d "$O/prof_probe" -o kt -- python "$R/tools/shard_scale_probe.py" "$a2" 6 fixed "${a3:-3}" > "$O/${a1}_probe_w$a2.log" 2>&1
        { grep -v amdgpu.ids "$O/${a1}_probe_w$a2.log" | tail -n 4; python "$R/tools/rocpd_stats.py" "$O/prof_probe/kt_results.db"; } > "$O/${a1}_shard_of_${a2}_kernel_stats.txt"
        head -n 40 "$O/${a1}_shard_of_${a2}_kernel_stats.txt"; rm -rf "$O/prof_probe" ) ;;
    c5stream) timeout 400 python tools/c5_stream.py $a2 > "$O/${a1}_c5_stream.json" 2> "$O/${a1}_c5_stream.err"; echo "rc=$?"; cut -c1-900 "$O/${a1}_c5_stream.json" ;;
    calib)   # known-byte-count kernels in K1's access patterns under FETCH_SIZE / WRITE_SIZE -> TAG_pmc_calibration.json
      ( cd /tmp
        for c in FETCH_SIZE WRITE_SIZE; do rm -rf "$O/cal_$c"; timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$O/cal_$c" -o pmc -- "$R/tools/pmc_calib" > "$O/cal_$c.log" 2>&1; done
        python "$R/tools/pmc_calib_json.py" "$O/cal_FETCH_SIZE" "$O/cal_WRITE_SIZE" "$a1" "$HEAD_ID" > "$O/${a1}_pmc_calibration.json"; cat "$O/${a1}_pmc_calibration.json"; rm -rf "$O/cal_FETCH_SIZE" "$O/cal_WRITE_SIZE" ) ;;
    box) { echo "# head $HEAD_ID"; rocm-smi --showclocks --showpower --showmaxpower --showperflevel --showmemorypartition --showcomputepartition 2>&1 | grep -v "^$\|====="; } > "$O/${a1}_box.txt" 2>&1; grep -i "sclk\|mclk\|fclk\|power\|partition" "$O/${a1}_box.txt" | head -n 14 ;;
    run) timeout 900 bash -c "$a1" 2>&1 | tail -n 40 ;;
    *) echo "unknown step $name" ;;
  esac
done
