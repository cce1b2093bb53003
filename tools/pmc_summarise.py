"""Summarise rocprofv3 --pmc csv output: per kernel name (cut to 60 chars),
mean of each counter over dispatches and mean duration; `@full` repeats the means over the FULL-SIZE dispatches only
(duration >= half of the longest one of that kernel), because a bench run also launches the kernel on small parity
samples and those pull a plain mean down."""
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
        full = defaultdict(lambda: defaultdict(list))
        rows = list(csv.DictReader(open(f)))
        longest = defaultdict(float)
        for r in rows:
            if r.get("End_Timestamp"):
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                longest[r["Kernel_Name"][:60]] = max(longest[r["Kernel_Name"][:60]], dur)
        for r in rows:
            name = r["Kernel_Name"][:60]
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                acc[name]["_dur_ns"].append(dur)
                if dur >= 0.5 * longest[name]:
                    full[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    full[name]["_dur_ns"].append(dur)
        print(f"== {os.path.basename(os.path.dirname(d))}")
        for name, cs in acc.items():
            if not (name.startswith("void qs::") or name.startswith("qs::")):
                continue
            parts = [f"{k}={sum(v)/len(v):.4g}(n={len(v)})" for k, v in sorted(cs.items())]
            print("  ", name, " ".join(parts))
            if name in full and len(full[name]["_dur_ns"]) != len(cs["_dur_ns"]):
                parts = [f"{k}={sum(v)/len(v):.4g}(n={len(v)})" for k, v in sorted(full[name].items())]
                print("  ", name, "@full", " ".join(parts))
