#!/usr/bin/env python3
"""Development aid: run one golden case through engine and oracle, list the records whose chosen row differs.
    python tools/debug_case.py cfg1_chr22 stat_default [accum]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import enginecase as ec  # noqa: E402
import goldencase as gc  # noqa: E402

case, run_name = sys.argv[1], sys.argv[2]
accum = int(sys.argv[3]) if len(sys.argv) > 3 else 1
run = gc.manifest_run(case, run_name)
p = gc.parse_opts(run["cmd"], run["opts"])
tm = gc.build_table_model(case, p["filter_field"], p["filter_name"])
header, rd = gc.load_reads(case, run["aln"])
rows = ec.table_from_model(tm)
eres, ores, hits = ec.run_both(rows, tm.chrom_size, tm.rep_len, len(tm.fams), len(tm.clas), p, gc.tid_map(header, tm, p["add_chr"]), rd,
                               batch_capacity=7001, accum=accum)
bad = np.nonzero(hits.astype(np.int64) != ores["hit_row"])[0]
print("records", len(hits), "mismatches", len(bad))
st, en = np.asarray(rows["start"], np.int64), np.asarray(rows["end"], np.int64)
for i in bad[:12]:
    pos, te, fl = int(rd["pos"][i]), int(rd["tmpend"][i]), int(rd["flag"][i])
    ext = p.get("extension", 150)
    qs, qe = (pos, pos + ext) if not fl & 16 else (max(te - ext, 0), te)
    cand = np.nonzero((st < qe) & (en > qs) & (np.asarray(rows["chrom"]) == gc.tid_map(header, tm, p["add_chr"])[int(rd["tid"][i])]))[0]
    print(f"rec {i} (in-batch {i % 7001}) tid {int(rd['tid'][i])} pos {pos} end {te} flag {fl} mapq {int(rd['mapq'][i])} -> gpu {int(hits[i])} oracle {int(ores['hit_row'][i])}")
    for c in cand:
        print(f"    row {c}: [{st[c]}, {en[c]}) ov {min(en[c], qe) - max(st[c], qs)}")
