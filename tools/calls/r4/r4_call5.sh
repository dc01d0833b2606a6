#!/bin/bash
cd "$GRAFT_REPO_ROOT"; O=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "Name:[[:space:]]*[A-Za-z0-9_]*" | awk '{print $2}' | sort -u > $O/c5_counters.txt
grep -i "ifetch\|icache\|SQC\|SQ_WAIT\|SQ_INST_LEVEL\|SQ_ACTIVE\|SQ_LEVEL\|STALL\|SQ_INSTS_SMEM\|SQ_BUSY" $O/c5_counters.txt | tr '\n' ' '; echo
wc -l $O/c5_counters.txt
R="$GRAFT_REPO_ROOT"
for k in tile team; do i=0
 for c in "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_IFETCH SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_INSTS_SALU" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES"; do
  i=$((i+1)); rm -rf $O/pq_$k$i
  SG_K1A=$k timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pq_$k$i -o pmc -- python $R/bench.py --config 3 --steps 6 --warmup 2 --profile-mode > $O/pq_$k$i.log 2>&1
 done
 python $R/tools/pmc_summary.py $O/pq_${k}1 $O/pq_${k}2 $O/pq_${k}3 2>/dev/null | grep -E "k1a|==" > $O/c5_pmc_$k.txt; cat $O/c5_pmc_$k.txt; rm -rf $O/pq_${k}1 $O/pq_${k}2 $O/pq_${k}3
done
