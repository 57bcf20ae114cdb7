#!/usr/bin/env python3
"""Scratch measurement (not the judged bench): device-resident batch through the engine, kernel time per path.

    python tools/quick_measure.py [n_rows] [n_reads] [accum,...]
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iteres_amd import engine as eng, synth  # noqa: E402


def main():
    n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
    n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    accums = [int(x) for x in (sys.argv[3] if len(sys.argv) > 3 else "1").split(",")]
    scale = n_rows / 5_500_000
    chroms = [(n, max(int(s * scale), 1000)) for n, s in synth.HG38_CHROMS]
    t0 = time.time()
    t = synth.make_table(20260101, chroms, n_rows, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    rl = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
    rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
    cs = np.array([s for _, s in chroms], np.int64)
    tid, pos, tmpend, mapq, f5 = synth.make_reads_soa(20260102, chroms, n_reads)
    print(f"gen {time.time() - t0:.1f}s rows={len(rows)} cov_len={int(rl.sum())}", flush=True)
    t0 = time.time()
    tab = eng.Table(rows, cs, rl, len(t.fams), len(t.clas))
    print(f"table build {time.time() - t0:.2f}s shift={tab.info.bin_shift} bytes={tab.info.table_bytes / 1e6:.1f}MB", flush=True)
    dev = torch.device("cuda:0")
    d = {k: torch.from_numpy(v).to(dev) for k, v in (("tid", tid), ("pos", pos), ("tmpend", tmpend), ("mapq", mapq), ("flag5", f5))}
    ptrs = {k: v.data_ptr() for k, v in d.items()}
    hit = torch.empty(n_reads, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    for accum in accums:
        e = eng.Engine(tab, dict(accum=accum), batch_capacity=n_reads)
        e.set_tidmap(list(range(len(chroms))))
        for rep in range(3):
            e.reset()
            e.submit_device(ptrs, n_reads)
            e.sync()
            st = e.stats()
            print(f"accum={accum} run{rep}: kernel_ms={st['kernel_ms']:.3f} hits={st['hits']} "
                  f"-> {n_reads / st['kernel_ms'] / 1e6:.2f} G reads/s", flush=True)
        e.reset()
        e.submit_device(ptrs, n_reads, hit_ptr=hit.data_ptr(), classify_only=True)
        e.sync()
        print(f"accum={accum} classify-only: kernel_ms={e.stats()['kernel_ms']:.3f}", flush=True)
        t0 = time.time()
        e.reset()
        e.submit_device(ptrs, n_reads)
        res = e.finish()
        print(f"finish {time.time() - t0:.3f}s cnt9={int(res['cnt'][9])} covsum={int(res['cov'].astype(np.uint64).sum())}", flush=True)
        e.close()
    tab.close()


if __name__ == "__main__":
    main()
