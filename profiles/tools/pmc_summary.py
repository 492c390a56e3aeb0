"""Summarise rocprofv3 --pmc counter_collection.csv files: mean counter value per kernel."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

for root in sys.argv[1:]:
    for f in sorted(Path(root).rglob("*counter_collection.csv")):
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            if not any(s in k for s in ("helm_patch", "helm_border", "ddh_wave_kernel", "ddh_block", "ddh_mfma")):
                continue
            print(k)
            for c, v in cs.items():
                print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
