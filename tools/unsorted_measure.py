#!/usr/bin/env python3
"""Scratch measurement: the same records in file order vs shuffled (name-sorted BAM): kernel time through the engine."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from iteres_amd import engine as eng, synth

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 20_000_000
chroms = synth.HG38_CHROMS
t = synth.make_table(20260101, chroms, 5_500_000, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
rl = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
cs = np.array([s for _, s in chroms], np.int64)
arrs = synth.make_reads_soa(20260102, chroms, n_reads)
tab = eng.Table(rows, cs, rl, len(t.fams), len(t.clas))
dev = torch.device("cuda:0")
perm = np.random.default_rng(5).permutation(n_reads)
for label, sel in (("sorted", None), ("shuffled", perm)):
    a = [x if sel is None else x[sel] for x in arrs]
    d = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dev) for k, v in zip(("tid", "pos", "tmpend", "mapq", "flag5"), a)}
    ptrs = {k: v.data_ptr() for k, v in d.items()}
    e = eng.Engine(tab, dict(), batch_capacity=n_reads)
    e.set_tidmap(list(range(len(chroms))))
    for rep in range(2):
        e.reset(); e.submit_device(ptrs, n_reads); e.sync()
        st = e.stats()
        print(label, rep, f"kernel_ms={st['kernel_ms']:.2f} -> {n_reads / st['kernel_ms'] / 1e6:.2f} G reads/s stages={[round(x, 3) for x in st['stage_ms']]}", flush=True)
    res = e.finish()
    print(label, "cnt9", int(res["cnt"][9]), "covsum", int(res["cov"].astype(np.uint64).sum()))
    e.close()
tab.close()
