#!/usr/bin/env python3
"""PCIe-inclusive rate of the hot path (Measurement): the record SoA sits in the engine's pinned host buffers
(14 B/record) and every batch crosses PCIe before the kernels run, two slots in flight — what the drop-in CLI's submit
loop does. Not the judged number (bench.py: records resident in HBM).

    python tools/pcie_rate.py [n_rows] [slot_records] [batches]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iteres_amd import engine as eng, synth  # noqa: E402


def main():
    n_rows = int(sys.argv[1]) if len(sys.argv) > 1 else 5_500_000
    cap = int(sys.argv[2]) if len(sys.argv) > 2 else 8_000_000
    batches = int(sys.argv[3]) if len(sys.argv) > 3 else 40
    scale = n_rows / 5_500_000
    chroms = [(n, max(int(s * scale), 1000)) for n, s in synth.HG38_CHROMS]
    t = synth.make_table(20260101, chroms, n_rows, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    rl = np.array([t.rep_len.get(n, 0) for n in t.names], np.uint32)
    rows = eng.make_rows(t.chrom, t.start, t.end, t.cons_start, t.cons_end, t.rep_name, t.fam_of_row, t.cla_of_row)
    tab = eng.Table(rows, np.array([s for _, s in chroms], np.int64), rl, len(t.fams), len(t.clas))
    tid, pos, tmpend, mapq, f5 = synth.make_reads_soa(20260102, chroms, 2 * cap)
    e = eng.Engine(tab, {}, batch_capacity=cap)
    e.set_tidmap(list(range(len(chroms))))
    L = eng.load()
    for s in (0, 1):
        b = e.staging(s)
        sl = slice(s * cap, (s + 1) * cap)
        b["tid"][:] = tid[sl]; b["pos"][:] = pos[sl]; b["tmpend"][:] = tmpend[sl]; b["mapq"][:] = mapq[sl]; b["flag5"][:] = f5[sl]
    for s in (0, 1):                                   # warm-up
        eng._chk(L.itx_engine_submit_slot(e._h, s, cap, 0, 0), "submit")
    for s in (0, 1):
        eng._chk(L.itx_engine_wait_slot(e._h, s), "wait")
    e.reset()
    t0 = time.perf_counter()
    for i in range(batches):
        s = i & 1
        if i >= 2:
            eng._chk(L.itx_engine_wait_slot(e._h, s), "wait")
        eng._chk(L.itx_engine_submit_slot(e._h, s, cap, 0, 0), "submit")
    for s in (0, 1):
        eng._chk(L.itx_engine_wait_slot(e._h, s), "wait")
    dt = time.perf_counter() - t0
    res = e.finish()
    n = cap * batches
    assert int(res["cnt"][0]) == n
    print(json.dumps({"records": n, "slot_records": cap, "seconds": round(dt, 4), "M_alignments_per_s": round(n / dt / 1e6, 1),
                      "host_to_device_GBps": round(14 * n / dt / 1e9, 2)}), flush=True)
    e.close()
    tab.close()


if __name__ == "__main__":
    main()
