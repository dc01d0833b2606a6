#!/bin/bash
# round 6, evidence set (one call, one box): the whole -m gpu suite, the C3 line with CPU baseline, end-to-end and churn legs, rocprofv3 kernel
# stats, FETCH / WRITE PMC passes (K1 + the whole window), SQ counters of the K1 kernels, MFMA-busy, phase stamps, the C2 line + kernel stats,
# the RCCL entry point at world = 1, one shard of 2 / 4 / 8 of C3, C5 as one shard of 8, the churn probe
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
T0=$(date +%s); lap() { echo "---- $1 at $(( $(date +%s) - T0 )) s"; }
TAG=${1:-r06_z}
tools/gpu.sh box:$TAG | head -n 4
tools/gpu.sh bench:${TAG}_c3 | cut -c1-300; lap bench3
tools/gpu.sh prof:$TAG:3 | head -n 24; lap prof3
tools/gpu.sh pmc:$TAG:3 | tail -n 16; lap pmc3
tools/gpu.sh sq:$TAG:3 | tail -n 30; lap sq3
tools/gpu.sh mfma:$TAG:3 | tail -n 8; lap mfma3
tools/gpu.sh stamps:3:SG_ABLATE=0x100 > /dev/null; cp $O/stamps_c3.log $O/${TAG}_stamps_c3.txt; lap stamps
tools/gpu.sh bench:${TAG}_c2:--config,2,--no-end-to-end,--cpu-seconds,2 | cut -c1-200; tools/gpu.sh prof:$TAG:2 | head -n 16; lap c2
SG_FORCE_SHARDED=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 30 --warmup 5 --no-cpu-baseline --verify > $O/${TAG}_sharded1.json 2> $O/${TAG}_sharded1.err; echo "sharded1 rc=$?"; lap sharded1
for w in 2 4 8; do
  python bench.py --config 3 --shard-of $w --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/${TAG}_c3_shard_of_$w.json 2> $O/${TAG}_c3_shard_of_$w.err; echo "shard-of $w rc=$?"
done; lap shards
python bench.py --config 5 --shard-of 8 --no-cpu-baseline --no-end-to-end --overlap-windows 0 > $O/${TAG}_c5_shard.json 2> $O/${TAG}_c5_shard.err; echo "c5 shard rc=$?"; lap c5
timeout 600 python tools/churn_probe.py 2>&1 | grep -v amdgpu.ids > $O/${TAG}_churn_probe.txt; tail -n 8 $O/${TAG}_churn_probe.txt; lap churn
tools/gpu.sh tests | tail -n 6; lap tests
python - $TAG <<'PY'
import json, sys
T = sys.argv[1]
def line(p):
    return json.loads(open(p).read().strip().splitlines()[-1])
j = line(f"gpurun_out/{T}_c3_bench.json")
print("C3", round(j["ms_per_step"] * 1e3, 1), j["per_step"], "frac", round(j["roofline"]["frac"], 4), j["roofline"]["pass_a_us"], j["roofline"]["pass_b_us"], j["roofline"]["traffic"], j["roofline"]["traffic_build_matches"])
print([(k["name"], k["us_per_window"]) for k in j["kernels"]], "alg GB/s", round(j["window_algorithmic_GBs"]), j["warm_windows"]["cold_ms_per_step"], j["overlapped"])
print("churn", j["warm_windows"]["churn"])
e = j["end_to_end"]; print("e2e", e["events_per_s"], e["frac_of_pcie_bound"], e["ring_full_retries"], e["registered_memory"]["events_per_s"], e["registered_memory"]["frac_of_pcie_bound"])
print("cpu", {k: j["cpu_baseline"].get(k) for k in ("value", "cores", "faithful_1t", "faithful_Nt", "lean_1t", "lean_Nt", "host_cpus")}, j["box"], j["effective_sclk_mhz"])
j = line(f"gpurun_out/{T}_c2_bench.json"); print("C2", round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], j["per_step"]["min_ms"], round(j["roofline"]["frac"], 4), [(k["name"], k["us_per_window"]) for k in j["kernels"]])
j = line(f"gpurun_out/{T}_sharded1.json"); print("sharded1", round(j["ms_per_step"] * 1e3, 1), j["comm_us_per_window"], j["rows_verified"], j["weak"]["ms_per_step"], j["kernels"])
for w in (2, 4, 8):
    j = line(f"gpurun_out/{T}_c3_shard_of_{w}.json"); print("shard-of", w, round(j["ms_per_step"] * 1e3, 1), j["per_step"]["median_ms"], j["config"]["events_per_window"], j["config"]["edges_per_window"], [(k["name"], k["us_per_window"]) for k in j["kernels"]], j["roofline"]["geometry"]["k1_narrow"], j["warm_windows"]["engine_keeps_state"])
j = line(f"gpurun_out/{T}_c5_shard.json"); print("c5 shard", round(j["ms_per_step"] * 1e3, 1), j["config"]["events_per_window"], j["config"]["edges_per_window"], round(j["roofline"]["frac"], 4), [(k["name"], k["us_per_window"]) for k in j["kernels"]])
PY
