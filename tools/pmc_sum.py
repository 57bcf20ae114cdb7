#!/usr/bin/env python3
"""Sums rocprofv3 --pmc counters per kernel: python tools/pmc_sum.py <dir with *_counter_collection.csv> [kernel name filter ...]"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    d = sys.argv[1]
    want = sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(float))
    launches = defaultdict(set)
    for fn in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"].split("(")[0]
                if want and not any(w in k for w in want):
                    continue
                acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
                launches[k].add(row["Dispatch_Id"])
    for k in sorted(acc):
        n = max(len(launches[k]), 1)
        print(f"{k}  ({n} launches; per launch)")
        for c in sorted(acc[k]):
            print(f"    {c:28s} {acc[k][c] / n:16.0f}")


if __name__ == "__main__":
    main()
