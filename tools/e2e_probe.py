#!/usr/bin/env python3
"""Host-feed probe: bench.end_to_end() of one BASELINE config for several feeder counts, pageable / registered caller memory,
overlapped (sg_flush_begin / sg_flush_end_view) or serial (sg_flush_window_view) window close.
usage: e2e_probe.py CONFIG 'FEEDERS/pinned|pageable/overlap|serial' ...   (environment, e.g. SG_STAGE_SLOTS, applies to the engine)"""
import os, sys, json, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from alaz_amd import engine, replay, weights

cfgno = int(sys.argv[1]); variants = sys.argv[2:] or ["8/pageable/overlap"]
c = replay.CONFIGS[cfgno]; seed = replay.SEED_BASE + cfgno
Ev = c["events"]; nb = int(os.environ.get("PROBE_NB", "2"))
topo = replay.make_topology(c["pods"], c["edges"], seed)
ev_all, labels = replay.make_events(topo, Ev * nb, seed)
a = types.SimpleNamespace(config=cfgno)
g = bench._engine_for(a, topo, labels, c, 0, 1, engine, weights)
g.ingest_bulk(ev_all[:Ev]); E = len(g.flush_window_view())
for v in variants:
    f, mem, mode = v.split("/")
    r = bench.end_to_end(g, ev_all, Ev, nb, int(f), E, pinned=(mem == "pinned"), serial=(mode == "serial"))
    print(f"[{v}] slots {os.environ.get('SG_STAGE_SLOTS', '16')}: {r['ms_per_window']:.2f} ms/window = {r['events_per_s'] / 1e9:.3f} G ev/s, frac of PCIe bound {r['frac_of_pcie_bound']}, "
          f"of the duplex bound {r['frac_of_duplex_bound']}, ring waits {r['ring_full_retries']}, rows {r.get('rows_by_window')}", flush=True)
g.close()
