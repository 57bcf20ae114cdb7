#!/usr/bin/env python3
"""Scratch measurement: itx_inflate_bgzf on a synthetic BAM (tools/mkbam.c), whole call and kernel-only, next to zlib
on one host core.   python tools/inflate_measure.py [n_reads=2000000] [seq_len=100] [repeat=3] [mkbam key=value options ...]
ITX_MEASURE_NOCHECK=1 skips the comparison with zlib (experiment builds whose pass 1 stores nothing)."""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iteres_amd import engine as eng, synth  # noqa: E402


def main():
    n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
    seq_len = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    repeat = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    mkopts = sys.argv[4:]
    mk = os.path.join(ROOT, "tools", "mkbam")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", mk, os.path.join(ROOT, "tools", "mkbam.c"), "-lz", "-ldl"])
    tmp = tempfile.mkdtemp(prefix="itx_inf_")
    synth.write_sizes(os.path.join(tmp, "chrom.sizes"), synth.HG38_CHROMS)
    subprocess.check_call([mk, os.path.join(tmp, "chrom.sizes"), str(n_reads), os.path.join(tmp, "reads.bam"), str(seq_len), "7"] + mkopts)
    comp = open(os.path.join(tmp, "reads.bam"), "rb").read()
    blocks = eng.index_bgzf(comp)
    total = int(blocks["usize"].astype(np.uint64).sum())
    print(f"mkbam options {mkopts}: {len(comp) / 1e6:.1f} MB compressed, {total / 1e6:.1f} MB inflated, {len(blocks)} blocks", flush=True)
    L = eng.load()
    h = eng.Inflater()
    cp = L.itx_pinned_alloc(len(comp) + 64)
    op = L.itx_pinned_alloc(total + 64)
    C.memmove(cp, comp, len(comp))
    status = np.zeros(len(blocks), np.uint8)
    for it in range(repeat):
        t0 = time.perf_counter()
        eng._chk(L.itx_inflate_bgzf(h._h, cp, len(comp), eng._p(blocks), len(blocks), op, total, eng._p(status)), "inflate")
        dt = time.perf_counter() - t0
        a, b = C.c_float(), C.c_float()
        L.itx_inflater_last_ms(h._h, C.byref(a), C.byref(b))
        print(f"call {it}: {dt * 1e3:.1f} ms -> {total / dt / 1e9:.2f} GB/s inflated (pass 1 {a.value:.2f} ms, pass 2 first group {b.value:.2f} ms), bad blocks {int((status != 0).sum())}", flush=True)
    # kernels only (nothing copied out): what the two passes do per launch of this many blocks
    L.itx_inflater_last_resolve_all_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    for it in range(2):
        eng._chk(L.itx_inflate_bgzf(h._h, cp, len(comp), eng._p(blocks), len(blocks), None, total, eng._p(status)), "inflate")
        a, b, c = C.c_float(), C.c_float(), C.c_float()
        L.itx_inflater_last_ms(h._h, C.byref(a), C.byref(b))
        L.itx_inflater_last_resolve_all_ms(h._h, C.byref(c))
    print(f"kernels only, {len(blocks)} blocks / {total / 1e6:.0f} MB per launch: pass 1 {a.value:.2f} ms = {total / a.value / 1e6:.1f} GB/s, pass 2 (all groups) {c.value:.2f} ms = {total / c.value / 1e6:.1f} GB/s of inflated bytes", flush=True)
    if os.environ.get("ITX_MEASURE_NOCHECK"):
        return
    out = np.ctypeslib.as_array(C.cast(op, C.POINTER(C.c_uint8)), shape=(total,))
    # every block against zlib (threads: zlib releases the GIL)
    from concurrent.futures import ThreadPoolExecutor
    t0 = time.perf_counter()

    def same(b):
        ref = zlib.decompress(comp[int(b["coff"]) + 18:int(b["coff"]) + int(b["csize"]) - 8], -15)
        return out[int(b["uoff"]):int(b["uoff"]) + int(b["usize"])].tobytes() == ref
    with ThreadPoolExecutor(16) as ex:
        oks = list(ex.map(same, blocks))
    dz = time.perf_counter() - t0
    print(f"all {len(blocks)} blocks equal zlib: {all(oks)} ({sum(oks)} equal); zlib on 16 threads {total / dz / 1e9:.2f} GB/s", flush=True)
    L.itx_pinned_free(cp)
    L.itx_pinned_free(op)
    h.close()


if __name__ == "__main__":
    main()
