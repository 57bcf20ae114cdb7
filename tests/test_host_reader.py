"""Host logic on CPU: the product's alignment decoder (iteres_amd/host/bamio.c: parallel BGZF inflate + record parse,
SAM text) against the independent Python readers of tests/refio.py, field by field, on every golden alignment file
and on stress files (many small BGZF blocks, records spanning blocks, odd batch sizes, truncation)."""
import os
import subprocess

import numpy as np
import pytest

import goldencase as gc
import refio
from iteres_amd import engine as eng, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "iteres_amd", "host")


@pytest.fixture(scope="module")
def dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bin") / "reader_dump")
    subprocess.check_call(["gcc", "-O2", "-g", "-fopenmp", "-std=gnu11", "-o", exe, os.path.join(HOST, "test", "reader_dump.c"),
                           os.path.join(HOST, "bamio.c"), os.path.join(HOST, "tables.c"), "-lz", "-ldl"])
    return exe


def run_dump(exe, path, is_sam, batch=4096, threads=4, chunk=None, piece=None):
    env = dict(os.environ, OMP_NUM_THREADS=str(threads))
    if chunk:
        env["ITX_BGZF_CHUNK"] = str(chunk)          # compressed bytes per read-ahead step: small = many buffer swaps
    if piece:
        env["ITX_HOP_PIECE"] = str(piece)           # bytes per piece of the parallel record search: small = many guesses
    pr = subprocess.run([exe, path, str(int(is_sam)), str(batch)], capture_output=True, text=True, env=env)
    assert pr.returncode == 0, pr.stderr
    hdr, recs, tail = [], [], None
    for line in pr.stdout.split("\n"):
        if line.startswith("@"):
            hdr.append(line.split("\t")[1])
        elif line.startswith("#"):
            tail = line
        elif line:
            recs.append(line.split("\t"))
    return hdr, recs, tail, pr.stderr


def check(hdr, recs, header, rd):
    assert hdr == [n for n, _ in header]
    assert len(recs) == len(rd["tid"])
    f5 = eng.flag5(rd["flag"])
    for i, r in enumerate(recs):
        want = [int(rd["tid"][i]), int(rd["pos"][i]), int(rd["tmpend"][i]), int(rd["mapq"][i]), int(f5[i]), int(rd["mpos"][i]), int(rd["isize"][i])]
        assert [int(x) for x in r[:7]] == want, (i, r, want)
        assert r[7] == rd["qname"][i]


@pytest.mark.parametrize("case", ["quirks", "mid", "cfg1_chr22", "manynames"])
def test_golden_alignment_files(case, dump, tmp_path):
    src = os.path.join(gc.GOLDEN, case, "in")
    for name in ("reads.bam", "reads.sam"):
        if not (os.path.exists(os.path.join(src, name)) or os.path.exists(os.path.join(src, name + ".gz"))):
            continue
        path = refio.materialise(src, name, str(tmp_path))
        header, rd = gc.load_reads(case, name)
        for batch in (4096, 777):
            hdr, recs, tail, _ = run_dump(dump, path, name.endswith(".sam"), batch)
            check(hdr, recs, header, rd)
        hdr, recs, tail, _ = run_dump(dump, path, name.endswith(".sam"), 4096, threads=3, piece=200)
        check(hdr, recs, header, rd)
        assert f"paired={int(bool((rd['flag'] & 1).any()))}" in tail


def test_record_search_recovers_from_wrong_guesses(dump, tmp_path):
    """Every record carries, in a byte-array tag, three well-formed little records in a row and then a length that leads
    off into the weeds: a piece that begins inside such a record guesses the decoy as its start. The stream-order check
    (bamio.c locate_records) must notice and walk those pieces again — the records that come out are the file's own."""
    import struct
    chroms = [("c1", 3_000_000)]
    r = synth.make_reads(78, chroms, 12_000, read_len=(30, 60), paired_frac=0.2)
    fake = struct.pack("<iiiIIiiii", 40, 0, 5, 2 | (30 << 8), 0, 0, -1, -1, 0) + b"a\0" + bytes(6)
    decoy = (fake * 3 + struct.pack("<i", 33) + bytes(range(40, 80))).hex()
    r.aux = [[f"ZZ:B:{decoy}"] if i % 3 else [] for i in range(len(r))]
    path = str(tmp_path / "decoy.bam")
    synth.write_bam(path, r, with_seq=True)
    header, rd = refio.read_bam(path)
    redone = 0
    for piece, threads in ((300, 4), (700, 3), (5000, 5)):
        env_t = dict(ITX_TIMING="1")
        os.environ.update(env_t)
        try:
            hdr, recs, tail, err = run_dump(dump, path, False, batch=5000, threads=threads, piece=piece)
        finally:
            os.environ.pop("ITX_TIMING")
        check(hdr, recs, header, rd)
        line = [l for l in err.split("\n") if "walked again" in l]
        assert line, err
        redone += int(line[0].split("pieces, ")[1].split(" walked")[0])
    assert redone > 0                       # the decoys did mislead some pieces, and the result is exact all the same


def test_many_small_blocks_and_truncation(dump, tmp_path):
    chroms = [("c1", 5_000_000), ("c2", 900_000)]
    r = synth.make_reads(77, chroms, 30_000, read_len=(30, 90), paired_frac=0.3, odd_cigar_frac=0.2)
    r.aux = [["NM:i:1", "XA:Z:c1,+100,30M,1;"] if i == 12345 else [] for i in range(len(r))]
    path = str(tmp_path / "small.bam")
    synth.write_bam(path, r, with_seq=True, block=700)             # thousands of tiny blocks: every record spans blocks
    header, rd = refio.read_bam(path)
    for threads in (1, 5):
        hdr, recs, tail, _ = run_dump(dump, path, False, batch=9999, threads=threads)
        check(hdr, recs, header, rd)
        assert "xa=1" in tail
    # the read-ahead thread with tiny steps: hundreds of buffer swaps, every one with a partial record carried over
    for chunk, batch in ((3000, 9999), (50_000, 1234), (1, 4096)):
        hdr, recs, tail, _ = run_dump(dump, path, False, batch=batch, threads=3, chunk=chunk)
        check(hdr, recs, header, rd)
    # the record search in pieces (bamio.c locate_records): pieces smaller than a record, of a few records, of many; with
    # and without sequence bytes (whose random content is where false guesses come from)
    for piece, threads, chunk in ((40, 4, None), (150, 3, None), (1000, 5, 50_000), (20_000, 2, None), (1 << 20, 4, None)):
        hdr, recs, tail, _ = run_dump(dump, path, False, batch=9999, threads=threads, chunk=chunk, piece=piece)
        check(hdr, recs, header, rd)
    # truncated in the middle of a block: a clean prefix of the records, no crash
    data = open(path, "rb").read()
    cut = str(tmp_path / "cut.bam")
    open(cut, "wb").write(data[: len(data) * 2 // 3 + 11])
    hdr, recs, tail, _ = run_dump(dump, cut, False, batch=4096)
    assert 0 < len(recs) < len(rd["tid"])
    hdr2, recs2, _, _ = run_dump(dump, cut, False, batch=4096, chunk=20_000)
    assert recs2 == recs
    hdr2, recs2, _, _ = run_dump(dump, cut, False, batch=4096, piece=500)
    assert recs2 == recs
    for i in (0, len(recs) // 2, len(recs) - 1):
        assert int(recs[i][1]) == int(rd["pos"][i]) and recs[i][7] == rd["qname"][i]
    # not a BAM at all
    bad = str(tmp_path / "bad.bam")
    open(bad, "wb").write(b"this is not a bam file" * 10)
    pr = subprocess.run([dump, bad, "0"], capture_output=True, text=True)
    assert pr.returncode == 1


# ---- split points of the multi-GPU shares (bamio.c: find_split) --------------------------------------------------------
@pytest.fixture(scope="module")
def split_dump(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bin") / "split_dump")
    subprocess.check_call(["gcc", "-O2", "-g", "-fopenmp", "-std=gnu11", "-o", exe, os.path.join(HOST, "test", "split_dump.c"),
                           os.path.join(HOST, "bamio.c"), os.path.join(HOST, "tables.c"), "-lz", "-ldl"])
    return exe


def _record_starts(path):
    """(blocks of the file, set of inflated offsets at which a record starts) by an independent walk"""
    import struct
    import zlib
    comp = open(path, "rb").read()
    blocks = eng.index_bgzf(comp)
    u = b"".join(zlib.decompress(comp[int(b["coff"]) + 18:int(b["coff"]) + int(b["csize"]) - 8], -15) for b in blocks)
    p = 4
    lt = struct.unpack_from("<i", u, p)[0]
    p += 4 + lt
    nref = struct.unpack_from("<i", u, p)[0]
    p += 4
    for _ in range(nref):
        ln = struct.unpack_from("<i", u, p)[0]
        p += 4 + ln + 4
    starts = set()
    while p + 4 <= len(u):
        starts.add(p)
        p += 4 + struct.unpack_from("<i", u, p)[0]
    return comp, blocks, starts


def _splits(exe, path, ats):
    out = subprocess.run([exe, path] + [str(a) for a in ats], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    return [tuple(int(x) for x in ln.split()) for ln in out.stdout.strip().split("\n")]


def test_split_points_are_record_starts(split_dump, tmp_path):
    """Where a rank's share of a BAM begins: for compressed offsets all over an ordinary file (short reads, a few long
    ones, blocks that start inside records) the point find_split gives is a block of the file, inside that block, at or
    after the offset asked for, and a true record start; offsets near the end give none."""
    chroms = [("c1", 30_000_000), ("c2", 8_000_000)]
    r = synth.make_reads(201, chroms, 60_000, read_len=(30, 160), paired_frac=0.2)
    path = str(tmp_path / "plain.bam")
    synth.write_bam(path, r, with_seq=True)
    comp, blocks, starts = _record_starts(path)
    uoff = {int(b["coff"]): (int(b["uoff"]), int(b["usize"]), int(b["csize"])) for b in blocks}
    rng = np.random.default_rng(3)
    ats = sorted(set([1, 100, len(comp) // 2, len(comp) - 29, len(comp) - 1] + [int(x) for x in rng.integers(1, len(comp), 60)]))
    found = 0
    for at, f, b, o, cs in _splits(split_dump, path, ats):
        if f == 0:
            assert at > int(blocks["coff"][-3])                      # only behind the last blocks that hold records
            continue
        assert f == 1 and b in uoff and b >= at and o < uoff[b][1] and cs == uoff[b][2]
        assert uoff[b][0] + o in starts, (at, b, o)
        found += 1
    assert found > 50


def test_split_points_can_be_fooled_by_decoys(split_dump, tmp_path):
    """The guess is only a guess: in a file that is nine tenths well-formed DECOY records (inside byte-array tags), the
    split points land on decoys — which is why the rank whose share ends there verifies the point, and why the job falls
    back to one rank when it does not hold (tests/test_cli_multi.py runs this very file through the command)."""
    import struct
    chroms = [("c1", 3_000_000)]
    r = synth.make_reads(92, chroms, 12_000, read_len=(30, 60))
    fake = struct.pack("<iiiIIiiii", 40, 0, 5, 2 | (30 << 8), 0, 0, -1, -1, 0) + b"a\0" + bytes(6)
    decoy = (fake * 40 + struct.pack("<i", 33) + bytes(range(40, 80))).hex()
    r.aux = [[f"ZZ:B:{decoy}"] for _ in range(len(r))]
    path = str(tmp_path / "decoy.bam")
    synth.write_bam(path, r, with_seq=True)
    comp, blocks, starts = _record_starts(path)
    uoff = {int(b["coff"]): int(b["uoff"]) for b in blocks}
    res = _splits(split_dump, path, [len(comp) * k // 4 for k in (1, 2, 3)])
    assert all(f == 1 for _, f, _, _, _ in res)
    assert any(uoff[b] + o not in starts for _, _, b, o, _ in res)


def test_split_points_with_long_records_of_mixed_lengths(split_dump, tmp_path):
    """Records of 60 - 120 KB between short ones (long reads): eight of them in a row are 0.5 - 1 MB of inflated bytes, more
    than any fixed margin of look-ahead. Every boundary of an 8-way split must still be decided — found (and a true record
    start) or "nothing starts behind it" — never given up on one side of a boundary while a later one holds: that made two
    ranks count the same records. The shares' union is then the file: a share ends where the next begins (same call, same
    answer), and once a boundary finds nothing no later one finds anything."""
    import struct
    import zlib
    rng = np.random.default_rng(77)
    raw = bytearray(b"BAM\1" + struct.pack("<i", 0) + struct.pack("<i", 1) + struct.pack("<i", 3) + b"c1\0" + struct.pack("<i", 200_000_000))
    pos = 0
    for i in range(1100):
        # the first 80 MB of the file hold long records only (no eight in a row fit a small margin for longer than any
        # give-up distance), the rest is mixed
        l_seq = int(rng.integers(60_000, 120_000)) if (i < 600 or rng.random() < 0.45) else int(rng.integers(50, 200))
        name = f"read{i}".encode() + b"\0"
        body = struct.pack("<iiIIiiii", 0, pos, (4681 << 16) | (30 << 8) | len(name), (0 << 16) | 1, l_seq, -1, -1, 0) + name + struct.pack("<I", l_seq << 4)
        body += bytes(rng.integers(0, 256, (l_seq + 1) // 2, dtype=np.uint8)) + bytes([30]) * l_seq
        raw += struct.pack("<i", len(body)) + body
        pos += int(rng.integers(1, 400))
    comp = bytearray()
    for off in range(0, len(raw), 0xff00):
        piece = bytes(raw[off:off + 0xff00])
        co = zlib.compressobj(1, zlib.DEFLATED, -15)
        d = co.compress(piece) + co.flush()
        comp += bytes.fromhex("1f8b08040000000000ff0600424302 00".replace(" ", "")) + struct.pack("<H", len(d) + 25) + d + struct.pack("<II", zlib.crc32(piece), len(piece))
    comp += bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")
    path = str(tmp_path / "long.bam")
    open(path, "wb").write(comp)
    _, blocks, starts = _record_starts(path)
    uoff = {int(b["coff"]): (int(b["uoff"]), int(b["usize"])) for b in blocks}
    n = 8
    res = _splits(split_dump, path, [len(comp) * k // n for k in range(1, n)] + [len(comp) - 40, len(comp) - 200_000])
    none_seen = False
    prev = -1
    for at, f, b, o, cs in sorted(res):
        assert f in (0, 1), f"boundary at {at} was given up"
        if f == 0:
            none_seen = True
            continue
        assert not none_seen, "a boundary behind one that found nothing found something"
        assert b in uoff and b >= at and o < uoff[b][1] and uoff[b][0] + o in starts, (at, b, o)
        assert uoff[b][0] + o >= prev
        prev = uoff[b][0] + o
    assert sum(f for _, f, _, _, _ in res) >= n - 2
