#!/usr/bin/env python3
"""Scratch measurement (not the judged bench): the partition path on N records resident in HBM, per-stage HIP-event
times, for one or more builds of the engine (tools/build_variant.sh) — each build in its own process (ITX_LIB).

    python tools/stream_measure.py [n_reads=500000000] [steps=10] [lib.so ...]
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one(n_reads, steps):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    torch.cuda.init()
    from iteres_amd import engine as eng, synth
    dev = torch.device("cuda", 0)
    chroms = synth.HG38_CHROMS
    tb = synth.make_table(20260101, chroms, 5_500_000, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    rep_len = np.array([tb.rep_len.get(n, 0) for n in tb.names], np.uint32)
    # ITX_MEASURE_SLOTS=<n>: consensus lengths scaled up to about n slots in all (the slot-space limit of the partition path);
    # ITX_MEASURE_ACCUM=atomic|partition: the accumulate path asked for
    want_slots = int(os.environ.get("ITX_MEASURE_SLOTS", "0"))
    if want_slots:
        rep_len = (rep_len.astype(np.float64) * (want_slots / float(rep_len.sum() + len(rep_len)))).astype(np.uint32)
    accum = {"atomic": eng.ACCUM_ATOMIC, "partition": eng.ACCUM_PARTITION}.get(os.environ.get("ITX_MEASURE_ACCUM", ""), eng.ACCUM_DEFAULT)
    rows = eng.make_rows(tb.chrom, tb.start, tb.end, tb.cons_start, tb.cons_end, tb.rep_name, tb.fam_of_row, tb.cla_of_row)
    cs = np.array([s for _, s in chroms], np.int64)
    table = eng.Table(rows, cs, rep_len, len(tb.fams), len(tb.clas))
    e = eng.Engine(table, {"accum": accum}, batch_capacity=n_reads)
    e.set_tidmap(list(range(len(chroms))))
    d = synth.make_reads_device(20260102, chroms, n_reads, dev)
    ptrs = {k: v.data_ptr() for k, v in d.items()}
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        e.submit_device(ptrs, n_reads, stream=st)
    e.sync()
    e.reset()
    for _ in range(steps):
        e.submit_device(ptrs, n_reads, stream=st)
    e.sync()
    s = e.stats()
    res = e.finish()
    out = {"lib": os.environ.get("ITX_LIB", "default"), "n": n_reads, "slots": int(rep_len.sum() + len(rep_len)), "kernel_ms_per_submit": s["kernel_ms"] / s["submits"], "stream_ms": s["stage_ms"][0] / s["submits"], "scatter_ms": s["stage_ms"][2] / s["submits"],
           "hist_ms": s["stage_ms"][3] / s["submits"], "keys": s["keys"] // s["submits"], "cnt9": int(res["cnt"][9]) // steps,
           "covsum": int(res["cov"].astype(np.uint64).sum()) // steps}
    print("RESULT " + json.dumps(out), flush=True)


def main():
    if os.environ.get("ITX_MEASURE_CHILD"):
        return one(int(sys.argv[1]), int(sys.argv[2]))
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000_000
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    libs = sys.argv[3:] or [""]
    for lib in libs:
        env = dict(os.environ, ITX_MEASURE_CHILD="1")
        if lib:
            env["ITX_LIB"] = os.path.abspath(lib)
        pr = subprocess.run([sys.executable, os.path.abspath(__file__), str(n), str(steps)], env=env, capture_output=True, text=True)
        got = [ln for ln in pr.stdout.split("\n") if ln.startswith("RESULT ")]
        print(got[0] if got else f"FAILED {lib}: {pr.stderr[-600:]}", flush=True)


if __name__ == "__main__":
    main()
