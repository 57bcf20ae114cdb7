#!/usr/bin/env python3
"""Digest of tools/profile_bench.sh's rocprofv3 output: for the python process of bench.py (the resident replay) the mean
duration of the k_stream / k_scatter / k_hist launches, and the PMC byte counts per k_stream launch (FETCH_SIZE doubled for
wide reads as MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE as it is; both counters are reported in KiB-free bytes
by rocprofv3's derived metric — the digest states the raw numbers too)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

root = sys.argv[1]


def rows(sub, pat):
    out = []
    for f in glob.glob(os.path.join(root, sub, "**", pat), recursive=True):
        out += [dict(r, _file=f) for r in csv.DictReader(open(f))]
    return out


res = {}
k = rows("stats", "*kernel_trace.csv")
by = defaultdict(list)
for r in k:
    name = r["Kernel_Name"].split("(")[0]
    by[(r.get("Process_Id") or r.get("Pid") or r["_file"], name)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
# the replay is the process with the longest k_stream launches
best = None
for (pid, name), v in by.items():
    if "k_stream" in name and (best is None or max(v) > max(by[best])):
        best = (pid, name)
if best:
    pid = best[0]
    res["replay_process"] = str(pid)
    for (p, name), v in by.items():
        if p == pid and any(x in name for x in ("k_stream", "k_scatter", "k_hist")):
            res[name.strip()] = {"launches": len(v), "mean_ms": round(sum(v) / len(v), 4), "min_ms": round(min(v), 4), "max_ms": round(max(v), 4)}
per_kernel = defaultdict(dict)
for sub, counter in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    c = rows(sub, "*counter_collection.csv")
    vals = defaultdict(list)
    for r in c:
        if r.get("Counter_Name") == counter and any(x in r.get("Kernel_Name", "") for x in ("k_stream", "k_scatter", "k_hist")):
            vals[r["Kernel_Name"].split("(")[0].strip()].append(float(r["Counter_Value"]))
    for name, v in vals.items():
        big = [x for x in v if x > 0.5 * max(v)]                  # the replay's launches (a small oracle-checked launch may ride along)
        per_kernel[name][counter] = {"launches": len(big), "mean_counter_value": sum(big) / len(big)}
res["pmc_per_kernel"] = {k: v for k, v in per_kernel.items()}
res["pmc_unit_note"] = "rocprofv3 derived counters, KB (1 KB = 1024 B) on this build; separate passes per counter"
# the EMIT variant of k_stream is the one that writes keys: the k_stream instantiation with the largest WRITE_SIZE
emit = None
for name, v in per_kernel.items():
    if "k_stream" in name and "WRITE_SIZE" in v and "FETCH_SIZE" in v and (emit is None or v["WRITE_SIZE"]["mean_counter_value"] > per_kernel[emit]["WRITE_SIZE"]["mean_counter_value"]):
        emit = name
if emit:
    f, w = per_kernel[emit]["FETCH_SIZE"]["mean_counter_value"], per_kernel[emit]["WRITE_SIZE"]["mean_counter_value"]
    res["k_stream_emit_kernel"] = emit
    res["FETCH_SIZE"] = per_kernel[emit]["FETCH_SIZE"]
    res["WRITE_SIZE"] = per_kernel[emit]["WRITE_SIZE"]
    res["k_stream_bytes_per_launch"] = int((2 * f + w) * 1024)
    res["how"] = "2 x FETCH_SIZE (gfx950 reports half the bytes of wide streaming reads) + WRITE_SIZE, counters in KB"
print(json.dumps(res, indent=1))
