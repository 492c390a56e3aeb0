"""Summarise rocprofv3 --pmc output (counter_collection.csv or the rocpd *_results.db): mean counter value per kernel."""
import csv
import sqlite3
import sys
from collections import defaultdict
from pathlib import Path

KEEP = ("helm_patch", "helm_lane", "helm_mfma", "helm_border", "op_patch", "op_border", "ddh_wave", "ddh_block", "ddh_mfma")

for root in sys.argv[1:]:
    acc = defaultdict(lambda: defaultdict(list))
    for f in sorted(Path(root).rglob("*counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in sorted(Path(root).rglob("*_results.db")):
        con = sqlite3.connect(f)
        for k, c, v in con.execute("select kernel_name, counter_name, value from counters_collection"):
            acc[k[:60]][c].append(float(v))
    for k, cs in acc.items():
        if not any(s in k for s in KEEP):
            continue
        print(k)
        for c, v in sorted(cs.items()):
            print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
