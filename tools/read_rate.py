#!/usr/bin/env python3
"""Scratch: how fast a file comes out of the page cache with N threads of pread into one reused buffer (what the reader thread of
host/bamio.c does per 128 MB chunk):  python tools/read_rate.py <file> [threads,...]"""
import os
import sys
import threading
import time


def run(path, n_thr, chunk=128 << 20):
    size = os.path.getsize(path)
    fd = os.open(path, os.O_RDONLY)
    buf = bytearray(chunk)
    mv = memoryview(buf)
    t0 = time.perf_counter()
    off = 0
    while off < size:
        want = min(chunk, size - off)
        per = (want + n_thr - 1) // n_thr

        def part(q):
            lo, hi = q * per, min(want, q * per + per)
            at = lo
            while at < hi:
                k = os.preadv(fd, [mv[at:hi]], off + at)
                if k <= 0:
                    break
                at += k
        ths = [threading.Thread(target=part, args=(q,)) for q in range(n_thr)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        off += want
    dt = time.perf_counter() - t0
    os.close(fd)
    return size / dt / 1e9, dt


if __name__ == "__main__":
    path = sys.argv[1]
    for n in [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,2,4,8,16").split(",")]:
        gbps, dt = run(path, n)
        print(f"{n} threads: {gbps:.1f} GB/s ({dt:.2f} s)", flush=True)
