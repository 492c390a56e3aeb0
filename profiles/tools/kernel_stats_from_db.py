"""rocprofv3 --kernel-trace --stats writes a rocpd database on this image; this turns its `kernels` view into the
per-kernel statistics CSV (Name, Calls, TotalDurationNs, AverageNs, Percentage, MinNs, MaxNs, StdDev).
usage: kernel_stats_from_db.py results.db out.csv"""
import csv
import sqlite3
import statistics
import sys

con = sqlite3.connect(sys.argv[1])
rows = {}
for name, s, e in con.execute("select name, start, end from kernels"):
    rows.setdefault(name, []).append(e - s)
total = sum(sum(v) for v in rows.values())
out = sorted(((n, len(v), sum(v), sum(v) / len(v), 100.0 * sum(v) / total, min(v), max(v), statistics.pstdev(v)) for n, v in rows.items()),
             key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in out:
        w.writerow([r[0], r[1], r[2], round(r[3], 3), round(r[4], 4), r[5], r[6], round(r[7], 3)])
for r in out[:10]:
    print(f"{r[0][:90]:90s} calls {r[1]:5d} avg {r[3] / 1e3:10.1f} us  {r[4]:6.2f} %")
