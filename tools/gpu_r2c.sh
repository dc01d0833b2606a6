#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L > gpurun_out/counters_list.txt 2>&1
grep -c "" gpurun_out/counters_list.txt
timeout 600 python tools/k1_sweep.py 3 "SG_NP=1024 SG_HT=2048" "SG_NP=1024 SG_HT=2048 SG_ABLATE=0x40" "SG_ABLATE=0x40" "SG_NP=512 SG_HT=2048" "SG_NP=512 SG_HT=2048 SG_ABLATE=0x40" "SG_NP=256 SG_HT=2048" > gpurun_out/sweep2_c3.log 2>&1
grep -v amdgpu.ids gpurun_out/sweep2_c3.log
cd /tmp
for c in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_sum" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
  n=$(echo $c | tr ' ' '_')
  SWEEP_STEPS=3 timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/pmc3_$n" -o pmc -- python "$GRAFT_REPO_ROOT/tools/k1_sweep.py" 3 "" > "$GRAFT_REPO_ROOT/gpurun_out/pmc3_$n.log" 2>&1
  echo "== $c rc=$?"
done
cd "$GRAFT_REPO_ROOT"
python - <<'PY'
import csv, glob, os, collections
for d in sorted(glob.glob('gpurun_out/pmc3_*')):
    if not os.path.isdir(d): continue
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        acc = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:28]
            if not (k.startswith('void k1a') or k.startswith('void k1b') or k.startswith('k1b') or k.startswith('k1a')): continue
            a = acc[(k, r['Counter_Name'])]; a[0] += float(r['Counter_Value']); a[1] += 1
        for (k, c), (v, n) in sorted(acc.items()):
            print(f"{k:30s} {c:28s} avg/launch {v / n:16.1f}  launches {n}")
PY
