"""GPU: itx_inflate_bgzf (one wavefront per BGZF block, iteres_amd/csrc/itx_inflate.hip) against zlib — the reference's
inflate_block (cussamtools/bgzf.c:367-397) is zlib's inflate, so zlib's bytes are the oracle here."""
import zlib

import numpy as np
import pytest

from iteres_amd import engine as eng, synth
from test_inflate_core import corpus, raw_deflate

pytestmark = pytest.mark.gpu


def member(data: bytes, level=6, strategy=zlib.Z_DEFAULT_STRATEGY) -> bytes:
    """one BGZF block (gzip member with the BC field) around a raw deflate stream of our choosing"""
    import struct
    comp = raw_deflate(data, level, strategy)
    bsize = len(comp) + 25
    assert bsize <= 65536
    return (b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00" + struct.pack("<H", bsize) + comp
            + struct.pack("<II", zlib.crc32(data) & 0xFFFFFFFF, len(data)))


@pytest.fixture(scope="module")
def inf():
    h = eng.Inflater()
    yield h
    h.close()


def test_every_block_type_level_and_alignment(inf):
    rng = np.random.default_rng(11)
    parts, want = [], []
    for data in corpus(rng):
        data = data[:60000]
        for level, strategy in ((0, zlib.Z_DEFAULT_STRATEGY), (1, zlib.Z_DEFAULT_STRATEGY), (6, zlib.Z_DEFAULT_STRATEGY), (9, zlib.Z_DEFAULT_STRATEGY),
                                (6, zlib.Z_FIXED), (6, zlib.Z_HUFFMAN_ONLY), (6, zlib.Z_RLE)):
            if level == 0 and len(data) > 65000:
                continue
            parts.append(member(data, level, strategy))      # members of every length: every alignment of input and output
            want.append(data)
    parts.append(synth.BGZF_EOF)
    want.append(b"")
    comp = b"".join(parts)
    out, status = inf.inflate(comp)
    assert len(status) == len(parts)
    assert (status == 0).all(), np.flatnonzero(status)[:10]
    assert out.tobytes() == b"".join(want)


def test_bam_like_chunk_many_blocks(inf, tmp_path):
    chroms = [("c1", 5_000_000), ("c2", 900_000)]
    r = synth.make_reads(91, chroms, 60_000, read_len=(50, 150), paired_frac=0.3, odd_cigar_frac=0.2)
    path = str(tmp_path / "x.bam")
    synth.write_bam(path, r, with_seq=True, level=6)
    comp = open(path, "rb").read()
    blocks = eng.index_bgzf(comp)
    assert len(blocks) > 100
    out, status = inf.inflate(comp, blocks)
    assert (status == 0).all()
    want = b"".join(zlib.decompress(comp[int(b["coff"]) + 18:int(b["coff"]) + int(b["csize"]) - 8], -15) for b in blocks)
    assert out.tobytes() == want
    # the same through small groups of blocks: offsets relative to a chunk that starts mid-file
    k = len(blocks) // 3
    sub = blocks[k:].copy()
    c0, u0 = int(sub["coff"][0]), int(sub["uoff"][0])
    sub["coff"] -= c0
    sub["uoff"] -= u0
    out2, status2 = inf.inflate(comp[c0:], sub)
    assert (status2 == 0).all() and out2.tobytes() == want[u0:]


def test_damaged_blocks_are_flagged_and_the_rest_is_right(inf):
    rng = np.random.default_rng(12)
    datas = [bytes(rng.integers(0, int(rng.choice([4, 30, 256])), int(rng.integers(100, 40000)), dtype=np.uint8)) for _ in range(300)]
    members = [bytearray(member(d, int(rng.choice([1, 6])))) for d in datas]
    hurt = set(int(x) for x in rng.choice(len(members), 60, replace=False))
    for i in hurt:
        m = members[i]
        for _ in range(3):
            m[int(rng.integers(18, len(m) - 8))] ^= 1 << int(rng.integers(0, 8))
    comp = b"".join(bytes(m) for m in members)
    out, status = inf.inflate(comp)
    blocks = eng.index_bgzf(comp)
    assert len(blocks) == len(members)
    flagged = 0
    for i, b in enumerate(blocks):
        got = out[int(b["uoff"]):int(b["uoff"]) + int(b["usize"])].tobytes()
        if i not in hurt:
            assert status[i] == 0 and got == datas[i], i
        elif status[i] == 0:
            # it decoded to the promised size: then zlib must read the same bytes out of the damaged stream
            d = zlib.decompressobj(-15)
            ref = d.decompress(comp[int(b["coff"]) + 18:int(b["coff"]) + int(b["csize"]) - 8])
            assert ref[:len(got)] == got, i
        else:
            flagged += 1
    assert flagged > 20
