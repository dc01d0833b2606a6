#!/usr/bin/env python3
"""Fold the FETCH_SIZE / WRITE_SIZE passes over tools/pmc_calib (known byte counts in K1's access patterns) into a calibration JSON:
for every kernel the bytes it really moved, what the counter reported (KiB -> bytes) and the factor reported / real.
usage: pmc_calib_json.py FETCH_DIR WRITE_DIR TAG HEAD"""
import csv, glob, json, os, sys
from collections import defaultdict

BYTES = 384 << 20
KNOWN = {   # kernel -> (bytes read, bytes written) per launch
    "cal_read16_stream": (BYTES, 0),
    "cal_read_pieces": ((BYTES // 3072) * 64, 0),
    "cal_write_runs64": (0, (BYTES // 3072) * 64),
    "cal_write_rows44": (0, (BYTES // 44) * 44),
    "cal_write16_stream": (0, BYTES),
}
PATTERN = {
    "cal_read16_stream": "pass A's event stream: 16 B per lane, coalesced",
    "cal_read_pieces": "pass B's piece reads: 4 lanes x 16 B per piece, pieces 3 KiB apart",
    "cal_write_runs64": "pass A's copy-out: 64-byte runs, not line-aligned, pieces 3 KiB apart",
    "cal_write_rows44": "pass B's compaction: 4 + 4 + 32 + 4 bytes per edge into four arrays, a workgroup's 1024 slots in mixed order",
    "cal_write16_stream": "plain coalesced 16 B per lane write",
}


def avg(d, counter):
    acc = defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                acc[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    fdir, wdir, tag, head = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "unknown"
    F, W = avg(fdir, "FETCH_SIZE"), avg(wdir, "WRITE_SIZE")
    out = {"round": tag, "git_head": head, "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- tools/pmc_calib",
           "note": "counter values are KiB; factor = reported bytes / bytes the kernel really moved (each launch touches memory nothing has cached)", "kernels": {}}
    for k, (rb, wb) in KNOWN.items():
        f = next((v for n, v in F.items() if k in n), None); w = next((v for n, v in W.items() if k in n), None)
        e = {"pattern": PATTERN[k], "read_bytes": rb, "written_bytes": wb}
        if f is not None: e["FETCH_SIZE_bytes"] = int(f * 1024); e["fetch_factor"] = round(f * 1024 / rb, 3) if rb else None
        if w is not None: e["WRITE_SIZE_bytes"] = int(w * 1024); e["write_factor"] = round(w * 1024 / wb, 3) if wb else None
        out["kernels"][k] = e
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
