#!/bin/bash
# round 5, call 24: sg_ingest sends a batch in 4 MiB pieces (the link works while the feeder copies the next piece): the host-fed parity tests,
# the default bench line (end_to_end), then the FETCH / WRITE passes again (servicegraph.hip changed: the hash bench.py compares covers it)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp SG_BENCH_CACHE=/tmp
mkdir -p gpurun_out
timeout 170 python -m pytest tests/test_gpu_parity.py -x -q -m gpu --durations=8 -k "config2_full_size or flush_begin_end or many_threads or concurrent_flushers or ingest_pinned or config3_full_size_row or config1_full or flush_window_view" 2>&1 | tail -n 14
timeout 200 python bench.py > gpurun_out/r05_piece_bench_c3.json 2> gpurun_out/r05_piece_bench_c3.err; echo "bench rc=$?"
python - <<'PY'
import json
j = json.load(open("gpurun_out/r05_piece_bench_c3.json"))
e = j["end_to_end"]; r = e.get("registered_memory", {})
print({k: j[k] for k in ("value", "ms_per_step")}, j["roofline"]["frac"], j["roofline"]["pass_a_us"])
print("pageable", e["events_per_s"], e["ms_per_window"], e["frac_of_pcie_bound"], e["ring_full_retries"], "| registered", r.get("events_per_s"), r.get("frac_of_pcie_bound"))
PY
tools/gpu.sh pmc:r05_zz:3 | tail -n 4
