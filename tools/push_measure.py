#!/usr/bin/env python3
"""Scratch: the device decoder's push pipeline alone (no file reading, no parse, no engine): the first GB_IN gigabytes of a
BAM held in page-locked memory, cut into chunks of CHUNK_MB, pushed with P pushes in flight.
    python tools/push_measure.py <bam> <gb_in> <chunk_mb,chunk_mb,...> <pushes,pushes,...>"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iteres_amd import engine as eng  # noqa: E402


def index_all(buf):
    """(coff, csize, usize) of every complete BGZF block of the buffer, vectorised walk is not possible: plain loop on memoryview"""
    out = []
    mv = memoryview(buf)
    off, n = 0, len(buf)
    while off + 18 <= n:
        bsize = (mv[off + 16] | (mv[off + 17] << 8)) + 1
        if off + bsize > n:
            break
        usize = int.from_bytes(bytes(mv[off + bsize - 4:off + bsize]), "little")
        out.append((off, bsize, usize))
        off += bsize
    return np.array(out, np.int64)


def main():
    path, gb = sys.argv[1], float(sys.argv[2])
    chunks = [int(x) for x in sys.argv[3].split(",")]
    pushes = [int(x) for x in sys.argv[4].split(",")]
    L = eng.load()
    L.itx_bamwin_push_begin.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
    L.itx_bamwin_push_end.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.POINTER(C.c_size_t)]
    L.itx_pinned_alloc.restype = C.c_void_p
    L.itx_pinned_alloc.argtypes = [C.c_size_t]
    L.itx_inflater_reserve.argtypes = [C.c_void_p, C.c_size_t, C.c_size_t, C.c_size_t, C.POINTER(C.c_int)]
    n_in = int(gb * 1e9)
    raw = np.fromfile(path, np.uint8, n_in)
    t0 = time.time()
    blk = index_all(raw)
    print(f"{len(blk)} blocks indexed in {time.time() - t0:.1f} s, {blk[:, 2].sum() / 1e9:.2f} GB inflated", flush=True)
    pin = L.itx_pinned_alloc(len(raw) + 64)
    C.memmove(pin, raw.ctypes.data, len(raw))
    for cm in chunks:
        # chunk boundaries: as many whole blocks as fit into cm MB, at most 16384 blocks / 1 GiB inflated
        cuts, a = [], 0
        while a < len(blk):
            b = a
            ub = 0
            while b < len(blk) and blk[b, 0] + blk[b, 1] - blk[a, 0] <= cm << 20 and b - a < 16384 and ub + blk[b, 2] <= (1 << 30) - (8 << 20):
                ub += blk[b, 2]
                b += 1
            cuts.append((a, b))
            a = b
        for P in pushes:
            h = eng.Inflater()
            nw = C.c_int(0)
            mb = max(b - a for a, b in cuts)
            rc = L.itx_inflater_reserve(h._h, (cm << 20) + (1 << 17), mb, min(mb * 65280, 1 << 30), C.byref(nw))
            assert rc == 0, L.itx_last_error()
            descs = []
            for a, b in cuts:
                d = np.zeros(b - a, eng.BGZF_BLOCK)
                d["coff"] = blk[a:b, 0] - blk[a, 0]
                d["csize"] = blk[a:b, 1]
                d["usize"] = blk[a:b, 2]
                d["uoff"] = np.concatenate([[0], np.cumsum(blk[a:b, 2])[:-1]])
                descs.append(d)
            status = np.zeros(mb + 64, np.uint8)
            n_new = C.c_size_t(0)
            for rep in range(2):
                t1 = time.perf_counter()
                begun = ended = 0
                while ended < len(cuts):
                    while begun < len(cuts) and begun - ended < P:
                        a, b = cuts[begun]
                        clen = int(blk[b - 1, 0] + blk[b - 1, 1] - blk[a, 0])
                        rc = L.itx_bamwin_push_begin(h._h, begun % nw.value, begun % 4, pin + int(blk[a, 0]), clen, eng._p(descs[begun]), b - a)
                        assert rc == 0, L.itx_last_error()
                        begun += 1
                    rc = L.itx_bamwin_push_end(h._h, ended % 4, eng._p(status), C.byref(n_new))
                    assert rc == 0, L.itx_last_error()
                    assert not status[: cuts[ended][1] - cuts[ended][0]].any()
                    ended += 1
                dt = time.perf_counter() - t1
            tot = blk[:, 2].sum()
            print(f"chunk {cm} MB ({len(cuts)} pushes of <= {mb} blocks), {P} in flight: {dt * 1e3:.0f} ms -> {tot / dt / 1e9:.1f} GB/s inflated, {dt / len(cuts) * 1e3:.1f} ms per push", flush=True)
            h.close()


if __name__ == "__main__":
    main()
