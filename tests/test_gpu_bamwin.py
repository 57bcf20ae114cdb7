"""GPU: BAM records located and parsed on the device (itx_bamwin_*, iteres_amd/csrc/itx_inflate.hip) against the
independent Python BAM reader of tests/refio.py — whole files in one window, files cut into many chunks (partial blocks
and partial records carried from window to window), decoy records that mislead the guesses, XA marks and raw bytes for
the side channels, a malformed length that ends the stream."""
import ctypes as C
import struct

import numpy as np
import pytest

import refio
from iteres_amd import engine as eng, synth

pytestmark = pytest.mark.gpu


class Win:
    """what the host reader does with the API: index blocks, push chunk by chunk, carry, parse, fetch"""

    def __init__(self):
        self.L = eng.load()
        self.h = eng.Inflater()

    def close(self):
        self.h.close()

    def read_all(self, comp: bytes, chunk: int, n_targets_hint=None):
        L, h = self.L, self.h._h
        blocks = eng.index_bgzf(comp)
        out = {k: [] for k in ("tid", "pos", "tmpend", "mapq", "flag5", "mpos", "isize", "xa", "qname")}
        w, first, hdr_done, n_targets, at, redo_total, malformed_any = 0, True, False, 0, 0, 0, False
        while at < len(blocks) and not malformed_any:
            c0 = int(blocks["coff"][at])
            j = at
            while j < len(blocks) and int(blocks["coff"][j]) + int(blocks["csize"][j]) - c0 <= max(chunk, int(blocks["csize"][at])):
                j += 1
            sub = blocks[at:j].copy()
            u0 = int(sub["uoff"][0])
            sub["coff"] -= c0
            sub["uoff"] -= u0
            c1 = int(blocks["coff"][j - 1]) + int(blocks["csize"][j - 1])
            cbuf = np.zeros(c1 - c0 + 16, np.uint8)
            cbuf[:c1 - c0] = np.frombuffer(comp[c0:c1], np.uint8)
            status = np.full(len(sub), 255, np.uint8)
            n_new = C.c_size_t()
            eng._chk(L.itx_bamwin_push(h, w, eng._p(cbuf), c1 - c0, eng._p(sub), len(sub), eng._p(status), C.byref(n_new)), "push")
            assert (status == 0).all()
            if not first:
                eng._chk(L.itx_bamwin_carry(h, 1 - w, w), "carry")
            first = False
            at = j
            if not hdr_done:
                avail = C.c_size_t()
                eng._chk(L.itx_bamwin_avail(h, w, C.byref(avail)), "avail")
                raw = np.zeros(avail.value, np.uint8)
                eng._chk(L.itx_bamwin_peek(h, w, 0, eng._p(raw), avail.value), "peek")
                b = raw.tobytes()
                try:
                    assert b[:4] == b"BAM\1"
                    l_text, = struct.unpack_from("<i", b, 4)
                    p = 8 + l_text
                    n_ref, = struct.unpack_from("<i", b, p)
                    p += 4
                    for _ in range(n_ref):
                        l_name, = struct.unpack_from("<i", b, p)
                        p += 4 + l_name + 4
                    assert p <= len(b)
                except (struct.error, AssertionError):
                    w ^= 1                      # header not complete yet: more input (tiny chunks)
                    continue
                n_targets = n_ref
                eng._chk(L.itx_bamwin_skip(h, w, p), "skip")
                hdr_done = True
            n_rec, mal, fl, redo = C.c_size_t(), C.c_int(), C.c_int(), C.c_size_t()
            eng._chk(L.itx_bamwin_parse(h, w, n_targets if n_targets_hint is None else n_targets_hint, C.byref(n_rec), C.byref(mal), C.byref(fl), C.byref(redo)),
                     "parse")
            redo_total += redo.value
            n = n_rec.value
            if n:
                arrs = {"tid": np.zeros(n, np.int32), "pos": np.zeros(n, np.int32), "tmpend": np.zeros(n, np.int32), "mapq": np.zeros(n, np.uint8),
                        "flag5": np.zeros(n, np.uint8), "mpos": np.zeros(n, np.int32), "isize": np.zeros(n, np.int32)}
                st = eng.Staging(*[arrs[k].ctypes.data for k in ("tid", "pos", "tmpend", "mapq", "flag5", "mpos", "isize")], None, n)
                off = np.zeros(n, np.uint32)
                xa = np.zeros(n, np.uint8)
                # in two fetches, to see the offsets argument work
                k = n // 3
                eng._chk(L.itx_bamwin_fetch(h, 0, k, C.byref(st), 0, eng._p(off), eng._p(xa)), "fetch")
                eng._chk(L.itx_bamwin_fetch(h, k, n - k, C.byref(st), k, eng._p(off[k:]), eng._p(xa[k:])), "fetch")
                for kk, v in arrs.items():
                    out[kk].append(v)
                out["xa"].append(xa)
                assert bool(fl.value & 2) == bool(xa.any()) and bool(fl.value & 1) == bool((arrs["flag5"] & 1).any())
                # read names through the raw bytes of the first and last few records
                for i in list(range(min(n, 3))) + list(range(max(n - 3, 0), n)):
                    rawrec = np.zeros(36 + 255, np.uint8)
                    eng._chk(L.itx_bamwin_bytes(h, int(off[i]), eng._p(rawrec), 36 + 255), "bytes")
                    lq = int(rawrec[12])
                    out["qname"].append((sum(len(a) for a in out["tid"]) - n + i, rawrec[36:36 + lq - 1].tobytes().decode()))
            malformed_any = bool(mal.value)
            w ^= 1
        res = {k: (np.concatenate(v) if v else np.zeros(0)) for k, v in out.items() if k != "qname"}
        res["qname"] = out["qname"]
        res["redo"] = redo_total
        res["malformed"] = malformed_any
        return res


@pytest.fixture(scope="module")
def win():
    w = Win()
    yield w
    w.close()


def check(res, rd):
    f5 = eng.flag5(rd["flag"])
    n = len(rd["tid"])
    assert len(res["tid"]) == n
    for k, want in (("tid", rd["tid"]), ("pos", rd["pos"]), ("tmpend", rd["tmpend"]), ("mapq", rd["mapq"]), ("flag5", f5), ("mpos", rd["mpos"]),
                    ("isize", rd["isize"])):
        assert np.array_equal(res[k].astype(np.int64), np.asarray(want).astype(np.int64)), k
    assert np.array_equal(res["xa"] != 0, np.array([x is not None for x in rd["xa"]]))
    for i, nm in res["qname"]:
        assert nm == rd["qname"][i]


def test_whole_file_and_many_chunks(win, tmp_path):
    chroms = [("c1", 5_000_000), ("c2", 900_000)]
    r = synth.make_reads(101, chroms, 50_000, read_len=(30, 150), paired_frac=0.3, odd_cigar_frac=0.2)
    rng = np.random.default_rng(102)
    r.aux = [["NM:i:1", "XA:Z:c1,+100,30M,1;"] if rng.random() < 0.01 else (["NM:i:0"] if rng.random() < 0.3 else []) for _ in range(len(r))]
    path = str(tmp_path / "a.bam")
    synth.write_bam(path, r, with_seq=True, level=6)
    header, rd = refio.read_bam(path)
    comp = open(path, "rb").read()
    for chunk in (1 << 30, 400_000, 70_000, 1):          # 1: one block per window
        res = win.read_all(comp, chunk)
        check(res, rd)
        assert not res["malformed"]
    # tiny blocks: every record straddles blocks, most windows end inside a record
    path2 = str(tmp_path / "b.bam")
    synth.write_bam(path2, r, with_seq=True, block=900)
    comp2 = open(path2, "rb").read()
    for chunk in (1 << 30, 20_000):
        check(win.read_all(comp2, chunk), rd)


def test_decoys_mislead_guesses_not_results(win, tmp_path):
    chroms = [("c1", 3_000_000)]
    r = synth.make_reads(103, chroms, 40_000, read_len=(30, 60), paired_frac=0.2)
    fake = struct.pack("<iiiIIiiii", 40, 0, 5, 2 | (30 << 8), 0, 0, -1, -1, 0) + b"a\0" + bytes(6)
    decoy = (fake * 3 + struct.pack("<i", 33) + bytes(range(40, 80))).hex()
    r.aux = [[f"ZZ:B:{decoy}"] if i % 3 else [] for i in range(len(r))]
    path = str(tmp_path / "decoy.bam")
    synth.write_bam(path, r, with_seq=True)
    header, rd = refio.read_bam(path)
    res = win.read_all(open(path, "rb").read(), 1 << 30)
    check(res, rd)
    assert res["redo"] > 0                      # some 16 KiB pieces did start inside a decoy-carrying record


def test_malformed_length_ends_the_stream(win, tmp_path):
    chroms = [("c1", 3_000_000)]
    r = synth.make_reads(104, chroms, 5_000, read_len=(30, 60))
    path = str(tmp_path / "m.bam")
    synth.write_bam(path, r, with_seq=False, eof=False)
    header, rd = refio.read_bam(path)
    # one more block whose "record" has a length below 32, then records that must never be seen
    bad = struct.pack("<i", 7) + bytes(60)
    comp = open(path, "rb").read() + synth.bgzf_block(bad) + synth.BGZF_EOF
    res = win.read_all(comp, 1 << 30)
    assert res["malformed"]
    check(res, rd)


def test_records_longer_than_a_piece_and_a_block(win, tmp_path):
    """Long reads: records of tens of kilobytes reach over several 16 KiB pieces (pieces without any record start) and over
    BGZF blocks; short ones in between."""
    chroms = [("c1", 50_000_000)]
    a = synth.make_reads(105, chroms, 300, read_len=(20_000, 45_000), odd_cigar_frac=0.5)
    b = synth.make_reads(106, chroms, 3000, read_len=(30, 200))
    # interleave by position: one sorted file
    import dataclasses
    order = np.lexsort((np.concatenate([a.pos, b.pos]), np.concatenate([a.tid, b.tid])))
    cat = lambda x, y: np.concatenate([x, y])[order]
    r = dataclasses.replace(a, tid=cat(a.tid, b.tid), pos=cat(a.pos, b.pos), flag=cat(a.flag, b.flag), mapq=cat(a.mapq, b.mapq), l_qseq=cat(a.l_qseq, b.l_qseq),
                            mtid=cat(a.mtid, b.mtid), mpos=cat(a.mpos, b.mpos), isize=cat(a.isize, b.isize),
                            cigars=[(a.cigars + b.cigars)[i] for i in order], qname=[(a.qname + [f"s{i}" for i in range(len(b))])[i] for i in order])
    path = str(tmp_path / "long.bam")
    synth.write_bam(path, r, with_seq=True)
    header, rd = refio.read_bam(path)
    comp = open(path, "rb").read()
    for chunk in (1 << 30, 300_000):
        check(win.read_all(comp, chunk), rd)


def test_record_longer_than_the_head_room_straddles_chunks(tmp_path):
    """A 5 MB record (a 3.3 M-base read) in a file the command pushes 300 KB at a time: its bytes pile up as the unconsumed tail
    of window after window until they exceed the 4 MiB of head room in front of a window's fresh bytes — the carry then moves
    the fresh bytes back instead of giving up (the reference and the host decoder read such files: bam.c:179-210 reallocs).
    Through the command: the device route, the host decoder and (when it travelled) the reference binary give the same files."""
    import dataclasses
    import filecmp
    import os
    import subprocess
    from iteres_amd import build
    lib, exe = build.build_all()
    chroms = [("c1", 50_000_000)]
    t = synth.make_table(110, chroms, 30_000, n_names=200, n_fams=20, n_clas=8)
    synth.write_sizes(str(tmp_path / "chrom.sizes"), chroms)
    synth.write_sizes(str(tmp_path / "rep.sizes"), t.rep_len.items())
    synth.write_rmsk(str(tmp_path / "rmsk.txt"), t)
    a = synth.make_reads(111, chroms, 1, read_len=(3_300_000, 3_300_001))
    b = synth.make_reads(112, chroms, 3000, read_len=(30, 200))
    order = np.lexsort((np.concatenate([a.pos, b.pos]), np.concatenate([a.tid, b.tid])))
    cat = lambda x, y: np.concatenate([x, y])[order]
    r = dataclasses.replace(a, tid=cat(a.tid, b.tid), pos=cat(a.pos, b.pos), flag=cat(a.flag, b.flag), mapq=cat(a.mapq, b.mapq), l_qseq=cat(a.l_qseq, b.l_qseq),
                            mtid=cat(a.mtid, b.mtid), mpos=cat(a.mpos, b.mpos), isize=cat(a.isize, b.isize),
                            cigars=[(a.cigars + b.cigars)[i] for i in order], qname=[(a.qname + [f"s{i}" for i in range(len(b))])[i] for i in order])
    synth.write_bam(str(tmp_path / "huge.bam"), r, with_seq=True)
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "iteres")
    runs = [("dev", exe, {"ITX_BGZF_CHUNK": "300000"}), ("host", exe, {"ITX_HOST_INFLATE": "1"})] + ([("ref", ref, {})] if os.path.exists(ref) else [])
    for name, prog, env in runs:
        out = tmp_path / name
        out.mkdir()
        pr = subprocess.run([prog, "stat", "-w", "-o", "out", str(tmp_path / "chrom.sizes"), str(tmp_path / "rep.sizes"), str(tmp_path / "rmsk.txt"), str(tmp_path / "huge.bam")],
                            cwd=out, capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert pr.returncode == 0, pr.stderr[-1500:]
    for name, _, _ in runs[1:]:
        for fn in sorted(os.listdir(tmp_path / "dev")):
            if not fn.endswith(".bigWig"):
                assert filecmp.cmp(tmp_path / "dev" / fn, tmp_path / name / fn, shallow=False), (name, fn)
