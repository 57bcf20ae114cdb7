#!/usr/bin/env python3
"""What a BAM's DEFLATE streams are made of: literals and matches per BGZF block, mean match length, Huffman decodes per
block, compressed and inflated bytes per read — the figures bench.py states next to the content it timed (a decoder's time
is set by the token mix, not by the byte count).

    python tools/bam_content.py <file.bam> [max_blocks=400]
    (as a module: content_stats(path, max_blocks) -> dict)

Tokens are counted by pass 1 of the product's own decoder built for the host (tests/inflate_host.cpp — the same source the
CPU suite fuzzes against zlib); every sampled block's bytes are also compared with zlib's."""
import ctypes as C
import json
import os
import struct
import subprocess
import sys
import tempfile
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_lib = None


def _host_lib():
    global _lib
    if _lib is None:
        so = os.path.join(tempfile.gettempdir(), f"libinflate_host_{os.getuid()}.so")
        src = os.path.join(ROOT, "tests", "inflate_host.cpp")
        core = os.path.join(ROOT, "iteres_amd", "csrc", "itx_inflate_core.h")
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(core)):
            subprocess.check_call(["g++", "-O2", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-DITXI_SIMPLE_IN", "-o", so, src])
        _lib = C.CDLL(so)
        _lib.itx_inflate_host.restype = C.c_int
        _lib.itx_inflate_host.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p]
    return _lib


def _blocks(path, want):
    """(offset, csize, usize) of up to `want` full-size blocks spread evenly over the file, found by walking BSIZE chains from
    `want` seek points (a chain of three headers in a row is a block boundary for every file this tool is pointed at)"""
    size = os.path.getsize(path)
    out = []
    with open(path, "rb") as f:
        for k in range(want):
            at = size * k // want
            f.seek(at)
            buf = f.read(1 << 18)
            i = 0
            while True:
                i = buf.find(b"\x1f\x8b\x08\x04", i)
                if i < 0 or i + 18 > len(buf):
                    i = -1
                    break
                ok, j = True, i
                for _ in range(3):
                    if j + 18 > len(buf) or buf[j:j + 4] != b"\x1f\x8b\x08\x04" or buf[j + 12:j + 14] != b"BC":
                        ok = j + 18 > len(buf) and _ > 0
                        break
                    j += struct.unpack_from("<H", buf, j + 16)[0] + 1
                if ok:
                    break
                i += 1
            if i < 0:
                continue
            cs = struct.unpack_from("<H", buf, i + 16)[0] + 1
            if i + cs > len(buf):
                continue
            us = struct.unpack_from("<I", buf, i + cs - 4)[0]
            out.append((at + i, buf[i:i + cs], us))
    seen, uniq = set(), []
    for o, b, u in out:
        if o not in seen:
            seen.add(o)
            uniq.append((o, b, u))
    return uniq, size


def _records_in(raw):
    """records that START in these inflated bytes cannot be counted without the chain; the mean record size of well-formed
    records found by scanning is enough: returns (n_records_found, bytes_they_span)"""
    n, span, p = 0, 0, 0
    L = len(raw)
    while p + 36 <= L:                                          # find a record start: block_size sane, l_qname ends in NUL
        bs, = struct.unpack_from("<i", raw, p)
        if 32 <= bs <= 4096 and p + 4 + bs <= L:
            lqn = raw[p + 12]
            if lqn and p + 36 + lqn <= L and raw[p + 36 + lqn - 1] == 0:
                q = p
                k = 0
                while q + 4 <= L:
                    b2, = struct.unpack_from("<i", raw, q)
                    if b2 < 32 or q + 4 + b2 > L:
                        break
                    q += 4 + b2
                    k += 1
                if k >= 3:
                    return k, q - p
        p += 1
    return n, span


def content_stats(path, max_blocks=400):
    L = _host_lib()
    blks, fsize = _blocks(path, max_blocks)
    n_lit = n_tok = usz = csz = recs = rec_bytes = 0
    nl, nt = C.c_uint32(), C.c_uint32()
    bad = 0
    for off, comp, us in blks:
        if us == 0:
            continue
        buf = np.zeros((len(comp) + 16 + 3) // 4 + 2, np.uint32)
        buf.view(np.uint8)[:len(comp)] = np.frombuffer(comp, np.uint8)
        out = np.zeros(us + 64, np.uint8)
        rc = L.itx_inflate_host(buf.ctypes.data, 18, len(comp) - 8, out.ctypes.data, 0, us, C.byref(nl), C.byref(nt))
        ref = zlib.decompress(comp[18:-8], -15)
        if rc != 0 or out[:us].tobytes() != ref:
            bad += 1
            continue
        n_lit += nl.value
        n_tok += nt.value
        usz += us
        csz += len(comp)
        k, sp = _records_in(ref)
        recs += k
        rec_bytes += sp
    nb = max(1, sum(1 for _, _, u in blks if u))
    rec_size = rec_bytes / recs if recs else 0.0
    ratio = csz / usz if usz else 0.0
    return {"blocks_sampled": nb, "blocks_differing_from_zlib": bad,
            "literals_per_block": round(n_lit / nb, 1), "matches_per_block": round(n_tok / nb, 1),
            "mean_match_bytes": round((usz - n_lit) / n_tok, 2) if n_tok else 0.0,
            "huffman_decodes_per_block": round((n_lit + 2 * n_tok) / nb, 1),
            "fraction_of_bytes_from_matches": round((usz - n_lit) / usz, 4) if usz else 0.0,
            "inflated_bytes_per_block": round(usz / nb, 1), "compressed_bytes_per_block": round(csz / nb, 1),
            "inflated_bytes_per_read": round(rec_size, 1), "compressed_bytes_per_read": round(rec_size * ratio, 1),
            "file_bytes": fsize, "inflated_bytes_estimate": int(fsize / ratio) if ratio else 0}


if __name__ == "__main__":
    print(json.dumps(content_stats(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 400), indent=1))
