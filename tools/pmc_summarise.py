"""Summarise rocprofv3 --pmc csv output: per kernel name (cut to 60 chars),
mean of each counter over dispatches and mean duration."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]
for d in sorted(glob.glob(os.path.join(root, "*/"))):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    for f in files:
        acc = defaultdict(lambda: defaultdict(list))
        for r in csv.DictReader(open(f)):
            name = r["Kernel_Name"][:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                acc[name]["_dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        print(f"== {os.path.basename(os.path.dirname(d))}")
        for name, cs in acc.items():
            if not (name.startswith("void qs::") or name.startswith("qs::")):
                continue
            parts = [f"{k}={sum(v)/len(v):.4g}(n={len(v)})" for k, v in sorted(cs.items())]
            print("  ", name, " ".join(parts))
