"""Summarise rocprofv3 output directories: per-kernel stats, and FETCH_SIZE / WRITE_SIZE per launch."""
import csv
import glob
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def short(name):
    name = name.replace("(anonymous namespace)", "anon").split("(")[0]
    for p in ("void ", "cap::anon::", "cap::pk::", "cap::"):
        name = name.replace(p, "")
    return name.strip()


for f in glob.glob(os.path.join(root, "trace", "**", "*kernel_stats.csv"), recursive=True):
    print("== kernel stats (rocprofv3 --kernel-trace --stats):", f)
    rows = list(csv.DictReader(open(f)))
    for r in rows[:25]:
        print(f"{short(r['Name'])[:60]:60s} calls {r['Calls']:>7s} total_ns {r['TotalDurationNs']:>14s} avg_ns {r['AverageNs']:>14s} pct {r['Percentage']}")
for kind in ("fetch", "write"):
    for f in glob.glob(os.path.join(root, f"pmc_{kind}", "**", "*counter_collection.csv"), recursive=True):
        agg = defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            agg[k][0] += float(r["Counter_Value"])
            agg[k][1] += 1
        print(f"== {kind.upper()}_SIZE per launch (raw counter units, rocprofv3 --pmc): {f}")
        for k, (v, c) in sorted(agg.items(), key=lambda kv: -kv[1][0])[:20]:
            print(f"{k[:60]:60s} launches {c:7d} sum {v:16.1f} per_launch {v / c:14.1f}")
