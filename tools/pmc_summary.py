#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --pmc counter from the counter_collection CSVs."""
import csv, glob, os, sys
from collections import defaultdict

for d in sys.argv[1:]:
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: defaultdict(list))
    for f in files:
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    print("==", d, len(files), "csv")
    for k, cs in sorted(acc.items()):
        for c, v in cs.items():
            print(f"{k:<62} {c:<12} n={len(v):>5} avg={sum(v)/len(v):>14.1f}")
