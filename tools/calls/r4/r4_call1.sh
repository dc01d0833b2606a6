#!/bin/bash
# round-4 call 1: issue-rate probe, then the C3 bench line (a) right away, (b) after 25 s of idle without settling, (c) after 25 s of idle with settling
cd "$GRAFT_REPO_ROOT"; O=gpurun_out; mkdir -p $O; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
tools/rate_probe > $O/r04_rate_probe.txt 2>&1; tail -n 45 $O/r04_rate_probe.txt
B="--no-cpu-baseline --no-end-to-end --overlap-windows 0"
python bench.py $B --settle-ms 0 > $O/c1_a.json 2> $O/c1_a.err; echo "a rc=$?"
sleep 25; python bench.py $B --settle-ms 0 > $O/c1_b.json 2> $O/c1_b.err; echo "b rc=$?"
sleep 25; python bench.py $B --settle-ms 400 > $O/c1_c.json 2> $O/c1_c.err; echo "c rc=$?"
python - <<'PY'
import json
for t in "abc":
    try:
        j = json.loads(open(f"gpurun_out/c1_{t}.json").read().strip().splitlines()[-1])
        print(t, round(j["ms_per_step"]*1e3,1), "us/window  K1a", round(j["roofline"]["pass_a_us"],1), "K1b", round(j["roofline"]["pass_b_us"],1), "frac", round(j["roofline"]["frac"],4), j["effective_sclk_mhz"], [ (k["name"], k["us_per_window"]) for k in j["kernels"]])
    except Exception as e: print(t, "ERR", e)
PY
rocm-smi --showclocks --showpower 2>&1 | grep -i "sclk\|power" | head -4
