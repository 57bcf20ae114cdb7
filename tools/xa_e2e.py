#!/usr/bin/env python3
"""Scratch / evidence: the whole command on a BAM with XA tags on a share of the reads (the veto is on by default,
stat.c:34,51), next to the same BAM run with -x (veto off), and optionally the reference on a sample.
    python tools/xa_e2e.py [n_reads=100000000] [xa_permille=400] [seq_len=100] [ref_reads=0]"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class A:
    pass


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000_000
    pm = int(sys.argv[2]) if len(sys.argv) > 2 else 400
    sl = int(sys.argv[3]) if len(sys.argv) > 3 else 100
    ref_n = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    a = A()
    a.reads, a.seq_len, a.rows, a.cpu_reads, a.workdir = 1000, sl, 5_500_000, 0, f"/tmp/itx_xa_{n}_{pm}_{sl}"
    wd, info = bench.ensure_inputs(a, 16)                      # the tables (and a token BAM)
    env = dict(os.environ, OMP_NUM_THREADS="16", ITX_TIMING="1", ITX_GPUS="1")
    bam = os.path.join(wd, "xa.bam")
    if not os.path.exists(bam):
        subprocess.check_call([bench.MKBAM, os.path.join(wd, "chrom.sizes"), str(n), bam, str(sl), "9", str(pm)], env=env)
    out = {"reads": n, "xa_permille": pm, "seq_len": sl, "bam_bytes": os.path.getsize(bam)}
    for name, head in (("veto_on", ["stat", "-w"]), ("veto_off_x", ["stat", "-w", "-x"])):
        walls = []
        for rep in range(2):
            args = head + ["-o", "out", os.path.join(wd, "chrom.sizes"), os.path.join(wd, "rep.sizes"), os.path.join(wd, "rmsk.txt"), bam]
            wall, rc, err, seen = bench.run_timed(bench.OURS, args, os.path.join(wd, name), env, (bench.SCAN_BEGIN, bench.SCAN_END))
            assert rc == 0, err[-800:]
            walls.append(round(wall, 3))
        out[name] = {"walls_s": walls, "M_reads_per_s": round(n / min(walls) / 1e6, 2), "phases": [ln for ln in err.split("\n") if ln.startswith("[itx timing] stream") or "BAM decode" in ln],
                     "report": open(os.path.join(wd, name, "out.iteres.report")).read().split("\n")[4]}
    if ref_n:
        rb = os.path.join(wd, "xa_ref.bam")
        subprocess.check_call([bench.MKBAM, os.path.join(wd, "chrom.sizes"), str(ref_n), rb, str(sl), "9", str(pm)], env=env)
        res = {}
        for name, exe in (("reference", bench.REF), ("drop_in", bench.OURS)):
            args = ["stat", "-w", "-o", "out", os.path.join(wd, "chrom.sizes"), os.path.join(wd, "rep.sizes"), os.path.join(wd, "rmsk.txt"), rb]
            wall, rc, err, seen = bench.run_timed(exe, args, os.path.join(wd, "s_" + name), env, ())
            res[name] = round(wall, 2)
        res["identical"] = all(open(os.path.join(wd, "s_reference", fn), "rb").read() == open(os.path.join(wd, "s_drop_in", fn), "rb").read() for fn in bench.TEXT_OUTPUTS)
        out["sample_vs_reference"] = dict(res, reads=ref_n)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
