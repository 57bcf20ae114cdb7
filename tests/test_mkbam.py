"""tools/mkbam.c — the generator of the measured inputs: every content / CIGAR / paired / pile-up mode
gives a well-formed, coordinate-sorted BAM with the stated properties; the default (legacy) mode's bytes are pinned so
that round-to-round figures stay comparable."""
import hashlib
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHROMS = [("chr1", 3_000_000), ("chr2", 1_200_000), ("chrX", 400_000)]


@pytest.fixture(scope="module")
def mk(tmp_path_factory):
    d = tmp_path_factory.mktemp("mkbam")
    exe = str(d / "mkbam")
    subprocess.check_call(["gcc", "-O2", "-fopenmp", "-Wall", "-Werror", "-o", exe, os.path.join(ROOT, "tools", "mkbam.c"), "-lz", "-ldl"])
    sizes = str(d / "chrom.sizes")
    with open(sizes, "w") as f:
        for n, s in CHROMS:
            f.write(f"{n}\t{s}\n")

    def run(n, *args, env=None):
        out = str(d / ("o_" + hashlib.md5(repr((n, args)).encode()).hexdigest()[:8] + ".bam"))
        subprocess.check_call([exe, sizes, str(n), out] + [str(a) for a in args], env=dict(os.environ, OMP_NUM_THREADS="4", **(env or {})), stderr=subprocess.DEVNULL)
        return out
    return run


def inflate_all(path):
    data = open(path, "rb").read()
    out, off = [], 0
    while off < len(data):
        assert data[off:off + 4] == b"\x1f\x8b\x08\x04" and data[off + 12:off + 14] == b"BC"
        bs = struct.unpack_from("<H", data, off + 16)[0] + 1
        raw = zlib.decompress(data[off + 18:off + bs - 8], -15)
        crc, isz = struct.unpack_from("<II", data, off + bs - 8)
        assert isz == len(raw) and crc == zlib.crc32(raw)
        out.append(raw)
        off += bs
    assert data[-28:] == bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    return b"".join(out)


def records(raw):
    assert raw[:4] == b"BAM\1"
    l_text, = struct.unpack_from("<i", raw, 4)
    off = 8 + l_text
    n_ref, = struct.unpack_from("<i", raw, off)
    off += 4
    for _ in range(n_ref):
        l_name, = struct.unpack_from("<i", raw, off)
        off += 8 + l_name
    assert n_ref == len(CHROMS)
    recs = []
    while off < len(raw):
        bs, tid, pos, bmn, fnc, l_seq, mtid, mpos, isize = struct.unpack_from("<iiiIIiiii", raw, off)
        l_qn, n_cig = bmn & 0xff, fnc & 0xffff
        p = off + 36
        name = raw[p:p + l_qn - 1]
        assert raw[p + l_qn - 1] == 0
        cig = struct.unpack_from(f"<{n_cig}I", raw, p + l_qn)
        q0 = p + l_qn + 4 * n_cig + (l_seq + 1) // 2
        recs.append(dict(tid=tid, pos=pos, mapq=(bmn >> 8) & 0xff, bin=bmn >> 16, flag=fnc >> 16, l_seq=l_seq, mtid=mtid, mpos=mpos, isize=isize, name=name,
                         cig=[(c & 15, c >> 4) for c in cig], seq=raw[p + l_qn + 4 * n_cig:q0], qual=raw[q0:q0 + l_seq], aux=raw[q0 + l_seq:off + 4 + bs]))
        off += 4 + bs
    assert off == len(raw)
    return recs


def is_sorted(recs):
    key = [(r["tid"], r["pos"]) for r in recs]
    return all(a <= b for a, b in zip(key, key[1:]))


def test_legacy_bytes_are_pinned(mk):
    raw = inflate_all(mk(30000, 60, 7, 50))
    recs = records(raw)
    assert len(recs) == 30000 and is_sorted(recs)
    # the inflated stream of round 2's generator for these arguments (the compressed bytes depend on the deflate library)
    assert hashlib.md5(raw).hexdigest() == "3a595259ff16d73000eea07251a01ed7"
    assert all(len(r["cig"]) == 1 and r["cig"][0][0] == 0 and 100 <= r["cig"][0][1] <= 150 for r in recs)
    assert all(s in (0x11, 0x22, 0x44, 0x88) for r in recs[:200] for s in r["seq"])


@pytest.mark.parametrize("content,n_values", [("hiseq", 40), ("novaseq", 4)])
def test_content_modes(mk, content, n_values):
    recs = records(inflate_all(mk(20000, 100, 5, 30, f"content={content}", "cigar=mixed")))
    assert len(recs) == 20000 and is_sorted(recs)
    quals = np.frombuffer(b"".join(r["qual"] for r in recs), np.uint8)
    vals = np.unique(quals)
    assert len(vals) <= n_values and vals.min() >= 2 and vals.max() <= 41
    if content == "hiseq":
        assert len(vals) >= 35
    bases = np.frombuffer(b"".join(r["seq"] for r in recs), np.uint8)
    hi, lo = bases >> 4, bases & 15
    assert set(np.unique(hi)) <= {1, 2, 4, 8} and set(np.unique(lo)) <= {1, 2, 4, 8, 15}
    # the two nibbles of a byte are independent: all 16 pairs occur about equally often
    pairs = np.bincount((hi[lo != 15] * 16 + lo[lo != 15]).astype(np.int64), minlength=256)
    seen = pairs[pairs > 0]
    assert len(seen) == 16 and seen.min() > 0.8 * seen.mean()
    mixed = [r for r in recs if len(r["cig"]) > 1]
    assert 0.03 * len(recs) < len(mixed) < 0.07 * len(recs)
    assert {op for r in mixed for op, _ in r["cig"]} == {0, 1, 2, 3, 4}
    for r in recs:
        assert sum(n for op, n in r["cig"] if op in (0, 1, 4)) == r["l_seq"] == 100          # M + I + S = query length
        assert r["name"].startswith(b"HS25_09078:") and r["name"].count(b":") == 4
    assert any(r["aux"].startswith(b"XAZ") for r in recs)


def test_paired_mode(mk):
    recs = records(inflate_all(mk(20001, 50, 9, "content=hiseq", "paired=1", "cigar=mixed")))
    assert len(recs) == 20001 and is_sorted(recs)
    by_name = {}
    for r in recs:
        by_name.setdefault((r["tid"], r["name"]), []).append(r)
    n_single = n_half = n_far = 0
    for (_, _), rs in by_name.items():
        if len(rs) == 1:
            assert rs[0]["flag"] & 1 == 0 and rs[0]["mtid"] == -1
            n_single += 1
            continue
        assert len(rs) == 2
        a, b = rs
        assert a["flag"] & 1 and b["flag"] & 1 and {a["flag"] & 0xc0, b["flag"] & 0xc0} == {0x40, 0x80}
        if a["flag"] & 0xc or b["flag"] & 0xc:
            m, u = (a, b) if a["flag"] & 8 else (b, a)
            assert m["flag"] & 8 and not m["flag"] & 4 and u["flag"] & 4 and m["pos"] == u["pos"] and not u["cig"]
            n_half += 1
            continue
        left, right = (a, b) if a["isize"] > 0 else (b, a)
        assert left["isize"] == -right["isize"] > 0 and left["mpos"] == right["pos"] and right["mpos"] == left["pos"]
        assert left["flag"] & 0x20 and not left["flag"] & 0x10 and right["flag"] & 0x10
        assert right["pos"] == max(left["pos"], left["pos"] + left["isize"] - 50) and 50 <= left["isize"] <= 700
        n_far += left["isize"] > 500
    assert n_single <= len(CHROMS) and 0.005 * len(by_name) < n_half < 0.04 * len(by_name) and n_far > 0
    sizes = [abs(r["isize"]) for r in recs if r["flag"] & 2 and r["isize"] > 0]
    assert 340 < np.mean(sizes) < 360 and 50 < np.std(sizes) < 70


def test_pileup_mode(mk):
    recs = records(inflate_all(mk(200000, 0, 3, "pileup=4")))
    assert len(recs) == 200000 and is_sorted(recs)
    pos = np.array([r["pos"] for r in recs if r["tid"] == 0])
    depth = np.bincount(pos // 64)
    assert depth.max() > 900 and np.median(depth) < 10                                    # a few loci thousands deep, the rest even


def test_unknown_option_is_refused(mk):
    with pytest.raises(subprocess.CalledProcessError):
        mk(10, "content=unknown")
