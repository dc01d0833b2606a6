#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes of `tools/gpu.sh pmc:TAG:CONFIG` into profiles-style JSON:
HBM bytes per launch of the two K1 kernels (bench.py reads it for roofline.traffic).

Corrections (MI355X_MICROARCH.md, HBM section): counter values are KiB; on gfx950 FETCH_SIZE reports
half the bytes of a wide coalesced 16 B/lane streaming read, so the event stream of pass A is doubled; pass B's
piece reads and every WRITE_SIZE are uncalibrated and taken as reported.  The JSON carries the git HEAD the
caller names and a hash of the kernel sources, so that bench.py can say whether the counters belong to the
build it is timing (`traffic_build_matches`)."""
import csv, glob, hashlib, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_src_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "alaz_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip", ".hpp")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]
from collections import defaultdict


def avg(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pick(m, prefix):
    """every kernel of the family (an engine that keeps warm-window state launches pass B twice per window — the warm attempt and the cold
    merge, one of which returns at once): their per-launch averages add up to the family's bytes per window"""
    tot, n = 0.0, 0
    for k, v in m.items():
        if prefix in k:
            tot += v[0]; n = max(n, v[1])
    return (tot, n)


def main():
    fdir, wdir, config, tag = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    head = sys.argv[5] if len(sys.argv) > 5 else "unknown"
    F, W = avg(fdir, "FETCH_SIZE"), avg(wdir, "WRITE_SIZE")
    # calibration (tools/pmc_calib under the same counters, profiles/*_pmc_calibration.json): reported / real bytes in K1's own access patterns
    cal = {"k1a_fetch": 0.5, "k1a_write": 1.0, "k1b_fetch": 1.0, "k1b_write": 1.0}; cal_src = None
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_calibration.json")))
    if cands:
        try:
            cj = json.load(open(cands[-1]))["kernels"]; cal_src = os.path.relpath(cands[-1], ROOT)
            cal = {"k1a_fetch": cj["cal_read16_stream"]["fetch_factor"], "k1a_write": cj["cal_write_runs64"]["write_factor"],
                   "k1b_fetch": cj["cal_read_pieces"]["fetch_factor"], "k1b_write": cj["cal_write_rows44"]["write_factor"]}
        except Exception:
            cal_src = None
    out = {"round": tag, "config": int(config), "git_head": head, "kernel_src_sha": kernel_src_sha(),
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --profile-mode",
           "unit_note": "counter values are KiB; k1a FETCH_SIZE doubled (gfx950 reports half of a coalesced 16 B/lane stream); "
                        "k1b FETCH_SIZE and all WRITE_SIZE taken as reported (uncalibrated)"}
    tot = tot_cal = 0.0
    for name, key, fmul in (("k1a", "k1a_", 2.0), ("k1b", "k1b_", 1.0)):
        f, nf = pick(F, key); w, nw = pick(W, key)
        b = (fmul * f + w) * 1024.0
        bc = (f / cal[name + "_fetch"] + w / cal[name + "_write"]) * 1024.0
        kname = next((k.split("(")[0] for k in F if key in k), key)
        out[name] = {"kernel": kname, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "launches": [nf, nw], "hbm_bytes": int(b),
                     "hbm_bytes_calibrated": int(bc)}
        tot += b; tot_cal += bc
    out["k1_total_hbm_bytes"] = int(tot)
    # the counters divided by what they report for a known byte count in the same access pattern (a write of a partial line is counted as
    # more than its bytes: 1.375 x for pass A's 64-byte runs, 1.63 x for pass B's rows).  Error bar: the two figures bracket the truth —
    # the calibration kernels write into untouched memory, the real kernels partly into lines that are completed by a neighbouring run.
    out["k1_total_hbm_bytes_calibrated"] = int(tot_cal)
    out["calibration"] = {"source": cal_src, "factors_reported_over_real": cal}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
