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
    for k, v in m.items():
        if prefix in k:
            return v
    return (0.0, 0)


def main():
    fdir, wdir, config, tag = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4]
    head = sys.argv[5] if len(sys.argv) > 5 else "unknown"
    F, W = avg(fdir, "FETCH_SIZE"), avg(wdir, "WRITE_SIZE")
    out = {"round": tag, "config": int(config), "git_head": head, "kernel_src_sha": kernel_src_sha(),
           "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python bench.py --profile-mode",
           "unit_note": "counter values are KiB; k1a FETCH_SIZE doubled (gfx950 reports half of a coalesced 16 B/lane stream); "
                        "k1b FETCH_SIZE and all WRITE_SIZE taken as reported (uncalibrated)"}
    tot = 0.0
    for name, key, fmul in (("k1a", "k1a_", 2.0), ("k1b", "k1b_", 1.0)):
        f, nf = pick(F, key); w, nw = pick(W, key)
        b = (fmul * f + w) * 1024.0
        kname = next((k.split("(")[0] for k in F if key in k), key)
        out[name] = {"kernel": kname, "FETCH_SIZE_KiB": round(f, 1), "WRITE_SIZE_KiB": round(w, 1), "launches": [nf, nw], "hbm_bytes": int(b)}
        tot += b
    out["k1_total_hbm_bytes"] = int(tot)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
