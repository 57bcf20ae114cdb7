#!/usr/bin/env python3
"""Summarise rocprofv3 CSV output (kernel stats / counter collection) per kernel: tools/pmc_summary.py <dir>..."""
import collections
import csv
import glob
import sys


def short(name):
    return name.split("(")[0].replace("void ", "")[:28]


for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
        print("==", f)
        for r in csv.DictReader(open(f)):
            print(f"  {short(r['Name']):30s} calls={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:10.1f} min_us={float(r['MinNs'])/1e3:10.1f} pct={r['Percentage']}")
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        print("==", f)
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in agg.items():
            print(f"  {k:30s}", {c: round(sum(x) / len(x)) for c, x in v.items()}, "n=", len(next(iter(v.values()))))
