#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (-d DIR -o NAME => DIR/NAME_results.db) as the
per-kernel table `rocprofv3 --stats` prints: calls, total, average, min, max (microseconds)."""
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                          "from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows) or 1.0
    lines = [f"{'kernel':<64} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}"]
    for r in rows:
        lines.append(f"{r[0][:64]:<64} {r[1]:>6} {r[2]:>12.1f} {r[3]:>10.2f} {r[4]:>10.2f} {r[5]:>10.2f} {100 * r[2] / tot:>6.2f}")
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    print(txt)


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
