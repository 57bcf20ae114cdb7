#!/usr/bin/env python3
"""Scratch (open question of round 3): does the host->device link run slower while threads copy a file out of the
page cache at the same time — what the reader of host/bamio.c does beside the decoder's copy stream?
    python tools/link_vs_reader.py <big file in the page cache> [chunk_mb=384] [copies=64] [reader threads=8]
Prints GB/s of the copies alone, of the reader alone, and of both together. --dry: no GPU (the copies go host to host)."""
import os
import sys
import threading
import time

import torch


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    dry = "--dry" in sys.argv
    path = args[0]
    chunk = int(args[1]) << 20 if len(args) > 1 else 384 << 20
    copies = int(args[2]) if len(args) > 2 else 64
    n_thr = int(args[3]) if len(args) > 3 else 8
    size = os.path.getsize(path)
    dev = "cpu" if dry else "cuda"
    src = [torch.empty(chunk, dtype=torch.uint8) for _ in range(2)]
    if not dry:
        src = [t.pin_memory() for t in src]
    for t in src:
        t.fill_(7)
    dst = [torch.empty(chunk, dtype=torch.uint8, device=dev) for _ in range(2)]
    rd_buf = [torch.empty(chunk, dtype=torch.uint8) for _ in range(2)]
    if not dry:
        rd_buf = [t.pin_memory() for t in rd_buf]
    for t in rd_buf:
        t.fill_(1)                                           # touched: placed where this process runs

    def do_copies(n):
        if dry:
            t0 = time.perf_counter()
            for k in range(n):
                dst[k & 1].copy_(src[k & 1])
            return n * chunk / (time.perf_counter() - t0) / 1e9
        st = torch.cuda.Stream()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with torch.cuda.stream(st):
            a.record(st)
            for k in range(n):
                dst[k & 1].copy_(src[k & 1], non_blocking=True)
            b.record(st)
        b.synchronize()
        return n * chunk / (a.elapsed_time(b) * 1e-3) / 1e9

    stop = threading.Event()
    read_bytes = [0]

    def reader():
        """chunk after chunk, each as n_thr slices read at once (host/bamio.c: io_main)"""
        fd = os.open(path, os.O_RDONLY)
        off, k = 0, 0
        per = (chunk + n_thr - 1) // n_thr
        while not stop.is_set():
            if off + chunk > size:
                off = 0
            mv = memoryview(rd_buf[k & 1].numpy())

            def part(q):
                lo, hi = q * per, min(q * per + per, chunk)
                at = lo
                while at < hi:
                    got = os.preadv(fd, [mv[at:hi]], off + at)
                    if got <= 0:
                        break
                    at += got
            ts = [threading.Thread(target=part, args=(q,)) for q in range(n_thr)]
            for t in ts:
                t.start()
            for t in ts:
                t.join()
            read_bytes[0] += chunk
            off += chunk
            k += 1
        os.close(fd)

    do_copies(4)                                              # warm
    alone = do_copies(copies)
    th = threading.Thread(target=reader)
    t0 = time.perf_counter()
    th.start()
    time.sleep(1.0)
    r_alone = read_bytes[0] / (time.perf_counter() - t0) / 1e9
    b0, t1 = read_bytes[0], time.perf_counter()
    together = do_copies(copies)
    r_together = (read_bytes[0] - b0) / (time.perf_counter() - t1) / 1e9
    stop.set()
    th.join()
    print(f"chunk {chunk >> 20} MB, {copies} copies, reader of {n_thr} threads on {path} ({size / 1e9:.1f} GB)")
    print(f"link alone      {alone:7.1f} GB/s")
    print(f"reader alone    {r_alone:7.1f} GB/s")
    print(f"link + reader   {together:7.1f} GB/s and {r_together:.1f} GB/s")


if __name__ == "__main__":
    main()
