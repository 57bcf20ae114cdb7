#!/usr/bin/env python3
"""CPU baseline with the REFERENCE ITSELF (oracle/_ref/iteres, compiled from /root/reference by oracle/Makefile),
next to the drop-in CLI, on files: the hg38-scale synthetic rmsk (5.5 M rows) and a coordinate-sorted BAM sample.

    python tools/ref_baseline.py [n_reads=3000000] [n_rows=5500000]

Prints one JSON line: wall seconds of `iteres stat -w` for both programs, M reads/s, and whether every output file
of the two runs is byte-identical. Test infrastructure: the reference binary is used here as baseline and checker."""
import filecmp
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iteres_amd import synth  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 3_000_000
    n_rows = int(sys.argv[2]) if len(sys.argv) > 2 else 5_500_000
    ref = os.path.join(ROOT, "oracle", "_ref", "iteres")
    ours = os.path.join(ROOT, "iteres_amd", "host", "iteres")
    assert os.path.exists(ref), "oracle/_ref/iteres missing (make -C oracle ref, in the build container)"
    scale = n_rows / 5_500_000
    chroms = synth.HG38_CHROMS if scale == 1 else [(n, max(int(s * scale), 1000)) for n, s in synth.HG38_CHROMS]
    t0 = time.time()
    tb = synth.make_table(20260101, chroms, n_rows, n_names=15000, n_fams=60, n_clas=20, overlap_frac=0.02)
    tid, pos, tmpend, mapq, f5 = synth.make_reads_soa(20260102, chroms, n_reads)
    tmp = tempfile.mkdtemp(prefix="itx_refbase_")
    synth.write_sizes(os.path.join(tmp, "chrom.sizes"), chroms)
    synth.write_sizes(os.path.join(tmp, "rep.sizes"), tb.rep_len.items())
    synth.write_rmsk(os.path.join(tmp, "rmsk.txt"), tb)
    rl = (tmpend - pos).astype(np.int32)
    reads = synth.Reads(list(chroms), tid, pos, np.where(f5 & 8, 16, 0).astype(np.uint16), mapq, rl, np.full(n_reads, -1, np.int32),
                        np.full(n_reads, -1, np.int32), np.zeros(n_reads, np.int32), [[("M", int(x))] for x in rl],
                        [f"r{i}" for i in range(n_reads)])
    synth.write_bam(os.path.join(tmp, "reads.bam"), reads, with_seq=False)
    gen_s = time.time() - t0
    out = {"n_reads": n_reads, "n_rows": n_rows, "gen_s": round(gen_s, 1), "host_cores": os.cpu_count()}
    args = ["stat", "-w", "-o", "out", os.path.join(tmp, "chrom.sizes"), os.path.join(tmp, "rep.sizes"), os.path.join(tmp, "rmsk.txt"),
            os.path.join(tmp, "reads.bam")]
    for name, exe in (("reference", ref), ("drop_in", ours)):
        wd = os.path.join(tmp, name)
        os.makedirs(wd)
        t1 = time.time()
        pr = subprocess.run([exe] + args, cwd=wd, capture_output=True, text=True)
        dt = time.time() - t1
        out[name] = {"rc": pr.returncode, "wall_s": round(dt, 2), "M_reads_per_s": round(n_reads / dt / 1e6, 4)}
        if pr.returncode != 0:
            out[name]["stderr_tail"] = pr.stderr[-500:]
    # the reference's scan phase alone: rerun it on an empty BAM to subtract table load + writers
    empty = synth.Reads(list(chroms), *(np.zeros(0, d) for d in (np.int32, np.int32, np.uint16, np.uint8, np.int32, np.int32, np.int32, np.int32)), [], [])
    synth.write_bam(os.path.join(tmp, "empty.bam"), empty, with_seq=False)
    wd = os.path.join(tmp, "ref_empty")
    os.makedirs(wd)
    t1 = time.time()
    subprocess.run([ref] + args[:-1] + [os.path.join(tmp, "empty.bam")], cwd=wd, capture_output=True, text=True)
    fixed = time.time() - t1
    scan = max(out["reference"]["wall_s"] - fixed, 1e-9)
    out["reference"]["fixed_s"] = round(fixed, 2)
    out["reference"]["scan_M_reads_per_s"] = round(n_reads / scan / 1e6, 4)
    same = {}
    for fn in sorted(os.listdir(os.path.join(tmp, "reference"))):
        if fn.endswith(".bigWig"):
            continue
        a, b = os.path.join(tmp, "reference", fn), os.path.join(tmp, "drop_in", fn)
        same[fn] = os.path.exists(b) and filecmp.cmp(a, b, shallow=False)
    out["files_identical"] = same
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
