"""Condense a rocprofv3 kernel_stats.csv into a short, committed summary
(kernel names cut to 100 chars).  Usage: condense_profile.py <kernel_stats.csv> <out.csv> [note]"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
rows = list(csv.DictReader(open(src)))
with open(dst, "w", newline="") as f:
    if note:
        f.write(f"# {note}\n")
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows[:15]:
        w.writerow([r["Name"][:100], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                    r["Percentage"], r["MinNs"], r["MaxNs"]])
